// Multifrontal block-sparse LL^T of the Schur complement on gfx950 (fp64).
//
// Replaces cuSOLVER's csrchol (ref: src/cholesky.hpp:97-155) — closed source there, written
// from scratch here.  The host (csrc/host/chol_symbolic.cpp) orders the 6x6-block graph,
// builds supernodes and a schedule of STAGES; a stage is one kernel launch in which every
// TASK (= one workgroup) processes a list of fronts sequentially:
//      front F = [ pivot block columns | boundary block rows | 1 rhs row ]   (dense, col-major)
//      1. extend-add the children's update matrices        (fixed child order: deterministic)
//      2. partial LL^T of the pivot columns, right-looking, 6-wide panels staged in LDS
//      3. the trailing part is the update matrix for the parent
// The right-hand side rides along as the last row of every front, so the forward
// substitution L y = b is a by-product of the factorisation (y ends up in the rhs row of the
// pivot columns); only the backward substitution needs its own top-down pass.
// A pivot <= 1e-14 (or NaN) raises *fail (ref: csrcholZeroPivot tol, src/cholesky.hpp:85).
#include "kernels.h"

namespace
{

constexpr int CBS = 256;
constexpr double PIVOT_TOL = 1e-14;

using cugo_k::CholPlanDev;

// ---------------------------------------------------------------- assembly -------------
__global__ __launch_bounds__(CBS) void k_assemble_blocks(CholPlanDev p, double* __restrict__ fronts,
                                                         const double* __restrict__ Hsc,
                                                         double lambda)
{
    const long idx = (long)blockIdx.x * CBS + threadIdx.x;
    if (idx >= 36L * p.n_hsc_blocks)
        return;
    const int k = (int)(idx / 36), t = (int)(idx % 36);
    const int r = t % 6, c = t / 6;
    const int f = p.blk_front[k];
    const long ld = 6L * p.nb[f] + 1;
    double* F = fronts + p.off[f];
    const int rb = p.blk_row[k], cb = p.blk_col[k];
    double v = Hsc[idx];
    if (rb == cb)
    {
        if (r < c)
            return; // keep the lower triangle of a diagonal block
        if (r == c)
            v += lambda;
        F[(6L * cb + c) * ld + 6L * rb + r] = v;
    }
    else if (!p.blk_trans[k])
        F[(6L * cb + c) * ld + 6L * rb + r] = v;
    else
        F[(6L * cb + r) * ld + 6L * rb + c] = v;
}

__global__ __launch_bounds__(CBS) void k_assemble_rhs(CholPlanDev p, double* __restrict__ fronts,
                                                      const double* __restrict__ bsc)
{
    const int j = blockIdx.x * CBS + threadIdx.x;
    if (j >= 6 * p.n)
        return;
    const int jb = j / 6, comp = j % 6;
    const int f = p.col_front[jb];
    const long ld = 6L * p.nb[f] + 1;
    const long lc = 6L * (jb - p.col0[f]) + comp;
    fronts[p.off[f] + lc * ld + (ld - 1)] = bsc[6L * p.perm[jb] + comp];
}

// ---------------------------------------------------------------- factor ---------------
__device__ void front_extend_add(const CholPlanDev& p, double* __restrict__ fronts, int f)
{
    const long ldp = 6L * p.nb[f] + 1;
    double* Fp = fronts + p.off[f];
    for (int ci = p.child_ptr[f]; ci < p.child_ptr[f + 1]; ci++)
    {
        const int c = p.child[ci];
        const int ncb = p.ncb[c], nbr = p.nb[c] - ncb;
        if (nbr == 0)
            continue;
        const long ldc = 6L * p.nb[c] + 1;
        const double* U = fronts + p.off[c] + (6L * ncb) * ldc + 6L * ncb; // (0,0) of update
        const int32_t* rel = p.rel + p.rel_ptr[c];
        const int nru = 6 * nbr + 1, ncu = 6 * nbr;
        for (int idx = threadIdx.x; idx < nru * ncu; idx += CBS)
        {
            const int j = idx / nru, i = idx % nru;
            if (i < j)
                continue;
            const long pj = 6L * rel[j / 6] + (j % 6);
            const long pi = (i == nru - 1) ? (ldp - 1) : 6L * rel[i / 6] + (i % 6);
            Fp[pj * ldp + pi] += U[(long)j * ldc + i];
        }
        __syncthreads(); // children are added one after the other
    }
}

__device__ void front_factor(const CholPlanDev& p, double* __restrict__ fronts, int f,
                             double* __restrict__ lds, int32_t* __restrict__ fail)
{
    const int ncb = p.ncb[f], nb = p.nb[f];
    const long ld = 6L * nb + 1;
    const int ncols = 6 * nb;
    double* F = fronts + p.off[f];
    double* D = lds;        // 36 : diagonal block / its factor
    double* Pn = lds + 40;  // panel rows [nrows_below][6]
    for (int kb = 0; kb < ncb; kb++)
    {
        const long j0 = 6L * kb;
        if (threadIdx.x < 36)
        {
            const int r = threadIdx.x % 6, c = threadIdx.x / 6;
            D[c * 6 + r] = F[(j0 + c) * ld + j0 + r];
        }
        __syncthreads();
        if (threadIdx.x == 0)
        {
            bool bad = false;
            for (int j = 0; j < 6; j++)
            {
                double d = D[j * 6 + j];
                for (int k = 0; k < j; k++)
                    d -= D[k * 6 + j] * D[k * 6 + j];
                if (!(d > PIVOT_TOL))
                {
                    bad = true;
                    d = 1.0; // keep going with finite numbers; the step is rejected anyway
                }
                d = sqrt(d);
                D[j * 6 + j] = d;
                const double inv = 1.0 / d;
                for (int i = j + 1; i < 6; i++)
                {
                    double s = D[j * 6 + i];
                    for (int k = 0; k < j; k++)
                        s -= D[k * 6 + i] * D[k * 6 + j];
                    D[j * 6 + i] = s * inv;
                }
            }
            if (bad)
                *fail = 1;
        }
        __syncthreads();
        if (threadIdx.x < 36)
        {
            const int r = threadIdx.x % 6, c = threadIdx.x / 6;
            if (r >= c)
                F[(j0 + c) * ld + j0 + r] = D[c * 6 + r];
        }
        // panel: rows below the diagonal block (including the rhs row)
        const int nbelow = (int)(ld - (j0 + 6));
        for (int i = threadIdx.x; i < nbelow; i += CBS)
        {
            const long row = j0 + 6 + i;
            double x[6];
#pragma unroll
            for (int c = 0; c < 6; c++)
            {
                double s = F[(j0 + c) * ld + row];
#pragma unroll
                for (int k = 0; k < 6; k++)
                    if (k < c)
                        s -= x[k] * D[k * 6 + c];
                x[c] = s / D[c * 6 + c];
            }
#pragma unroll
            for (int c = 0; c < 6; c++)
            {
                F[(j0 + c) * ld + row] = x[c];
                Pn[i * 6 + c] = x[c];
            }
        }
        __syncthreads();
        // trailing update: F[i,j] -= Pn[i]·Pn[j] for j in [j0+6, ncols), i >= j (and rhs row)
        const int nc2 = ncols - (int)(j0 + 6);
        const long total = (long)nbelow * nc2;
        for (long idx = threadIdx.x; idx < total; idx += CBS)
        {
            const int j = (int)(idx / nbelow), i = (int)(idx % nbelow);
            if (i < j)
                continue;
            const double* a = Pn + i * 6;
            const double* b = Pn + j * 6;
            const double s = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] +
                             a[5] * b[5];
            F[(j0 + 6 + j) * ld + (j0 + 6 + i)] -= s;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(CBS) void k_factor_stage(CholPlanDev p, double* __restrict__ fronts,
                                                      int task0, int32_t* __restrict__ fail)
{
    extern __shared__ double lds[];
    const int task = task0 + blockIdx.x;
    for (int fi = p.task_ptr[task]; fi < p.task_ptr[task + 1]; fi++)
    {
        const int f = p.task_fronts[fi];
        front_extend_add(p, fronts, f);
        __syncthreads();
        front_factor(p, fronts, f, lds, fail);
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------- backward -------------
__global__ __launch_bounds__(CBS) void k_backward_stage(CholPlanDev p,
                                                        const double* __restrict__ fronts,
                                                        int task0, double* __restrict__ xnew,
                                                        double* __restrict__ xout)
{
    extern __shared__ double lds[];
    const int task = task0 + blockIdx.x;
    for (int fi = p.task_ptr[task + 1] - 1; fi >= p.task_ptr[task]; fi--)
    {
        const int f = p.task_fronts[fi];
        const int ncb = p.ncb[f], nb = p.nb[f];
        const long ld = 6L * nb + 1;
        const int ncs = 6 * ncb, nrs = 6 * (nb - ncb);
        const double* F = fronts + p.off[f];
        double* xr = lds;        // nrs
        double* vs = lds + nrs;  // ncs
        const int32_t* rows = p.rows + p.rows_ptr[f];
        for (int i = threadIdx.x; i < nrs; i += CBS)
            xr[i] = xnew[6L * rows[i / 6] + (i % 6)];
        __syncthreads();
        for (int j = threadIdx.x; j < ncs; j += CBS)
        {
            const double* col = F + (long)j * ld;
            double s = col[ld - 1]; // y_j
            for (int i = 0; i < nrs; i++)
                s -= col[ncs + i] * xr[i];
            vs[j] = s;
        }
        __syncthreads();
        for (int j = ncs - 1; j >= 0; j--)
        {
            if (threadIdx.x == 0)
                vs[j] /= F[(long)j * ld + j];
            __syncthreads();
            const double xj = vs[j];
            for (int t = threadIdx.x; t < j; t += CBS)
                vs[t] -= F[(long)t * ld + j] * xj;
            __syncthreads();
        }
        const int c0 = p.col0[f];
        for (int j = threadIdx.x; j < ncs; j += CBS)
        {
            const int jb = j / 6, comp = j % 6;
            const double v = vs[j];
            xnew[6L * (c0 + jb) + comp] = v;
            xout[6L * p.perm[c0 + jb] + comp] = v;
        }
        __threadfence_block();
        __syncthreads();
    }
}

} // namespace

namespace cugo_k
{

size_t chol_lds_factor_bytes(long ld_max) { return (size_t)(40 + 6 * ld_max) * sizeof(double); }
size_t chol_lds_backward_bytes(long ld_max) { return (size_t)(ld_max + 8) * sizeof(double); }

static void ensure_lds(const void* fn, size_t bytes)
{
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

void launch_chol_assemble(hipStream_t s, const CholPlanDev& p, double* d_fronts,
                          size_t front_doubles, const double* d_Hsc, double lambda,
                          const double* d_bsc)
{
    (void)hipMemsetAsync(d_fronts, 0, front_doubles * sizeof(double), s);
    const long n = 36L * p.n_hsc_blocks;
    if (n > 0)
        hipLaunchKernelGGL(k_assemble_blocks, dim3((unsigned)((n + CBS - 1) / CBS)), dim3(CBS), 0, s,
                           p, d_fronts, d_Hsc, lambda);
    if (p.n > 0)
        hipLaunchKernelGGL(k_assemble_rhs, dim3((6 * p.n + CBS - 1) / CBS), dim3(CBS), 0, s, p,
                           d_fronts, d_bsc);
}

void launch_chol_factor_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                              int ntasks, size_t lds_bytes, int32_t* d_fail)
{
    if (ntasks <= 0)
        return;
    ensure_lds(reinterpret_cast<const void*>(k_factor_stage), lds_bytes);
    hipLaunchKernelGGL(k_factor_stage, dim3(ntasks), dim3(CBS), lds_bytes, s, p, d_fronts, task0,
                       d_fail);
}

void launch_chol_backward_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                                int ntasks, size_t lds_bytes, double* d_xnew, double* d_x)
{
    if (ntasks <= 0)
        return;
    ensure_lds(reinterpret_cast<const void*>(k_backward_stage), lds_bytes);
    hipLaunchKernelGGL(k_backward_stage, dim3(ntasks), dim3(CBS), lds_bytes, s, p, d_fronts, task0,
                       d_xnew, d_x);
}

} // namespace cugo_k
