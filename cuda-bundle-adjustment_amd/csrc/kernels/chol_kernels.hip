// Multifrontal block-sparse LL^T of the Schur complement on gfx950 (fp64).
//
// Replaces cuSOLVER's csrchol (ref: src/cholesky.hpp:97-155) — closed source there, written
// from scratch here.  The host (csrc/host/chol_symbolic.cpp) orders the 6x6-block graph
// (nested dissection), builds supernodes ("fronts", pivot width <= 96 scalars so that L11
// lives in LDS) and a schedule of STAGES:
//
//   front F = [ pivot block columns | boundary block rows | 1 rhs row ]   dense, column-major
//
//   stage 0 (optional)  : whole bottom subtrees, one workgroup per subtree, fronts in postorder
//   upper stages        : one etree level per stage, four batched kernels per level so that a
//                         big front is spread over many workgroups (256 CUs / 8 XCDs):
//        extend-add  children update matrices -> parent front (fixed child order)
//        potrf       L11 = chol(F11) in LDS                       (1 workgroup / front)
//        trsm        L21 = F21 L11^-T, 64-row tiles, L11 + tile in LDS
//        syrk        U = F22 - L21 L21^T, 64x64 tiles on the f64 matrix cores
//                    (v_mfma_f64_16x16x4_f64), lower tiles only
//
// The right-hand side rides along as the last row of every front, so L y = b is a by-product
// of the factorisation (y ends in the rhs row of the pivot columns); only the backward
// substitution needs its own top-down pass.  All sums have a fixed order: bit-reproducible.
// A pivot <= 1e-14 (or NaN) raises *fail (ref: csrcholZeroPivot tol, src/cholesky.hpp:85).
#include <algorithm>

#include "kernels.h"

namespace
{

constexpr int CBS = 256;
constexpr double PIVOT_TOL = 1e-14;
constexpr int TR = 64; // trsm / syrk tile edge

using cugo_k::CholPlanDev;
typedef double double4_t __attribute__((ext_vector_type(4)));

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

// ---------------------------------------------------------------- device building blocks
// children -> parent, restricted to parent block columns [cb0, cb1); children one after the
// other (barrier in between) so every parent entry is summed in child order.  One wave per
// child column, lanes stride the rows: coalesced reads of U, 8-B scattered RMW on the parent.
__device__ void dev_extend_add(const CholPlanDev& p, double* __restrict__ fronts, int f, int cb0,
                               int cb1)
{
    const long ldp = 6L * p.nb[f] + 1;
    double* Fp = fronts + p.off[f];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int ci = p.child_ptr[f]; ci < p.child_ptr[f + 1]; ci++)
    {
        const int c = p.child[ci];
        const int ncb = p.ncb[c], nbr = p.nb[c] - ncb;
        const int32_t* rel = p.rel + p.rel_ptr[c];
        // child update block columns whose parent column falls in [cb0, cb1) (rel is ascending)
        int jlo = 0;
        while (jlo < nbr && rel[jlo] < cb0)
            jlo++;
        int jhi = jlo;
        while (jhi < nbr && rel[jhi] < cb1)
            jhi++;
        if (jhi > jlo)
        {
            const long ldc = 6L * p.nb[c] + 1;
            const double* U = fronts + p.off[c] + (6L * ncb) * ldc + 6L * ncb; // (0,0) of update
            const int nru = 6 * nbr + 1;
            for (int j = 6 * jlo + wv; j < 6 * jhi; j += CBS / 64)
            {
                const long pj = 6L * rel[j / 6] + (j % 6);
                const double* ucol = U + (long)j * ldc;
                double* pcol = Fp + pj * ldp;
                for (int i = j + lane; i < nru; i += 64)
                {
                    const long pi = (i == nru - 1) ? (ldp - 1) : 6L * rel[i / 6] + (i % 6);
                    pcol[pi] += ucol[i];
                }
            }
        }
        __syncthreads();
    }
}

// in-register Cholesky of the 6x6 diagonal block at (j0,j0) of an LDS matrix (one thread)
__device__ __forceinline__ bool chol6_lds(double* __restrict__ Ls, int lds, int j0)
{
    double a[6][6];
#pragma unroll
    for (int c = 0; c < 6; c++)
#pragma unroll
        for (int r = 0; r < 6; r++)
            a[r][c] = (r >= c) ? Ls[(j0 + c) * lds + j0 + r] : 0.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 6; j++)
    {
        double d = a[j][j];
#pragma unroll
        for (int k = 0; k < 6; k++)
            if (k < j)
                d -= a[j][k] * a[j][k];
        if (!(d > PIVOT_TOL))
        {
            bad = true;
            d = 1.0; // keep finite numbers flowing; the LM step is rejected anyway
        }
        d = sqrt(d);
        a[j][j] = d;
        const double inv = 1.0 / d;
#pragma unroll
        for (int i = 0; i < 6; i++)
            if (i > j)
            {
                double s = a[i][j];
#pragma unroll
                for (int k = 0; k < 6; k++)
                    if (k < j)
                        s -= a[i][k] * a[j][k];
                a[i][j] = s * inv;
            }
    }
#pragma unroll
    for (int c = 0; c < 6; c++)
#pragma unroll
        for (int r = 0; r < 6; r++)
            if (r >= c)
                Ls[(j0 + c) * lds + j0 + r] = a[r][c];
    return bad;
}

// L11 = chol(F11) in LDS (Ls: nc x nc, leading dimension nc+1), blocked by 6 columns.
// On return Ls holds L11 (lower) and F11 in global memory is overwritten with it.
__device__ void dev_potrf(double* __restrict__ F, long ld, int nc, double* __restrict__ Ls,
                          int32_t* __restrict__ fail)
{
    const int lds = nc + 1;
    for (int idx = threadIdx.x; idx < nc * nc; idx += CBS)
    {
        const int r = idx % nc, c = idx / nc;
        Ls[c * lds + r] = (r >= c) ? F[(long)c * ld + r] : 0.0;
    }
    __syncthreads();
    for (int j0 = 0; j0 < nc; j0 += 6)
    {
        if (threadIdx.x == 0)
        {
            if (chol6_lds(Ls, lds, j0))
                *fail = 1;
        }
        __syncthreads();
        const int m = nc - (j0 + 6); // rows below the diagonal block inside F11
        for (int i = threadIdx.x; i < m; i += CBS)
        { // panel rows: x L_D^T = a
            const int row = j0 + 6 + i;
            double x[6];
#pragma unroll
            for (int c = 0; c < 6; c++)
            {
                double s = Ls[(j0 + c) * lds + row];
#pragma unroll
                for (int k = 0; k < 6; k++)
                    if (k < c)
                        s -= x[k] * Ls[(j0 + k) * lds + j0 + c];
                x[c] = s / Ls[(j0 + c) * lds + j0 + c];
            }
#pragma unroll
            for (int c = 0; c < 6; c++)
                Ls[(j0 + c) * lds + row] = x[c];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < m * m; idx += CBS)
        { // trailing update inside F11 (lower part)
            const int c = j0 + 6 + idx / m, r = j0 + 6 + idx % m;
            if (r < c)
                continue;
            double s = 0;
#pragma unroll
            for (int k = 0; k < 6; k++)
                s += Ls[(j0 + k) * lds + r] * Ls[(j0 + k) * lds + c];
            Ls[c * lds + r] -= s;
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < nc * nc; idx += CBS)
    {
        const int r = idx % nc, c = idx / nc;
        if (r >= c)
            F[(long)c * ld + r] = Ls[c * lds + r];
    }
}

__device__ void dev_load_l11(const double* __restrict__ F, long ld, int nc, double* __restrict__ Ls)
{
    const int lds = nc + 1;
    for (int idx = threadIdx.x; idx < nc * nc; idx += CBS)
    {
        const int r = idx % nc, c = idx / nc;
        Ls[c * lds + r] = (r >= c) ? F[(long)c * ld + r] : 0.0;
    }
}

// rows [row0, row0+nrows) (absolute scalar rows, nrows <= TR) of F21: X L11^T = B.
// Ls holds L11; Bt is a TR x (nc+1) LDS tile.  4 adjacent lanes share one row, so a row
// never leaves its wave: LDS operations of one wave execute in order, a wave-level fence is
// all the synchronisation the column loop needs.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ void dev_trsm_tile(double* __restrict__ F, long ld, int nc, long row0, int nrows,
                              const double* __restrict__ Ls, double* __restrict__ Bt)
{
    const int lds = nc + 1;
    for (int idx = threadIdx.x; idx < nrows * nc; idx += CBS)
    {
        const int r = idx % nrows, c = idx / nrows;
        Bt[r * lds + c] = F[(long)c * ld + row0 + r];
    }
    __syncthreads();
    const int r = threadIdx.x >> 2, g = threadIdx.x & 3; // row in tile, column group
    if (r < nrows)
    {
        double* row = Bt + r * lds;
        for (int j0 = 0; j0 < nc; j0 += 6)
        {
            // every lane of the row solves the 6x6 system redundantly (same inputs, same result)
            double x[6];
#pragma unroll
            for (int c = 0; c < 6; c++)
            {
                double s = row[j0 + c];
#pragma unroll
                for (int k = 0; k < 6; k++)
                    if (k < c)
                        s -= x[k] * Ls[(j0 + k) * lds + j0 + c];
                x[c] = s / Ls[(j0 + c) * lds + j0 + c];
            }
            wave_lds_sync(); // all 4 lanes have read the old values
            if (g == 0)
            {
#pragma unroll
                for (int c = 0; c < 6; c++)
                    row[j0 + c] = x[c];
            }
            for (int c = j0 + 6 + g; c < nc; c += 4)
            {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 6; k++)
                    s += x[k] * Ls[(j0 + k) * lds + c];
                row[c] -= s;
            }
            wave_lds_sync(); // updates visible to the row's other lanes
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < nrows * nc; idx += CBS)
    {
        const int rr = idx % nrows, c = idx / nrows;
        F[(long)c * ld + row0 + rr] = Bt[rr * lds + c];
    }
    __syncthreads();
}

// U(ti,tj) -= L21(ti rows) L21(tj rows)^T for one 64x64 tile on the f64 matrix cores.
// L21 = F[ncs.., 0..ncs) (column-major, rows contiguous); U = F[ncs.., ncs..).
// nt = trailing rows (boundary + rhs row), nrs = trailing columns.
// Both 64 x ncs row panels are staged in LDS by the whole workgroup in one pass (all loads in
// flight at once: the K loop then runs from LDS without touching memory latency), k-major
// with an 80-double stride so the two 16-lane halves of a ds_read_b64 hit different banks.
// Wave w owns the 16 U-columns [64 tj + 16 w, +16) and all 64 U-rows (4 accumulators).
// MFMA operand map (v_mfma_f64_16x16x4_f64): lane l supplies A[m = l&15][k = l>>4] and
// B[k = l>>4][n = l&15]; result reg q holds D[m = (l>>4) + 4q][n = l&15].  With m = U column
// and n = U row the 16 lanes l&15 hit consecutive rows of one column: 128-B segments.
constexpr int PST = 80; // LDS panel stride (doubles) per k
__device__ void dev_syrk_tile(double* __restrict__ F, long ld, int ncs, int nt, int nrs, int ti,
                              int tj, double* __restrict__ lds)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ln = lane & 15, lk = lane >> 4;
    const double* L21 = F + ncs; // element (row i, col k) = L21[k*ld + i]
    double* Pi = lds;                 // rows of tile ti : Pi[k*PST + r]
    double* Pj = lds + ncs * PST;     // rows of tile tj (aliases Pi on diagonal tiles)
    const bool diag = ti == tj;
    if (diag)
        Pj = Pi;
    for (int idx = threadIdx.x; idx < 64 * ncs; idx += CBS)
    {
        const int r = idx & 63, k = idx >> 6;
        const int gi = 64 * ti + r;
        Pi[k * PST + r] = gi < nt ? L21[(long)k * ld + gi] : 0.0;
        if (!diag)
        {
            const int gj = 64 * tj + r;
            Pj[k * PST + r] = gj < nt ? L21[(long)k * ld + gj] : 0.0;
        }
    }
    __syncthreads();
    double4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
        acc[t] = double4_t{0, 0, 0, 0};
    const int kend = (ncs + 3) & ~3;
    for (int k0 = 0; k0 < kend; k0 += 4)
    {
        const int k = k0 + lk;
        const bool kok = k < ncs;
        const double a = kok ? Pj[k * PST + 16 * w + ln] : 0.0;
#pragma unroll
        for (int t = 0; t < 4; t++)
        {
            const double b = kok ? Pi[k * PST + 16 * t + ln] : 0.0;
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
    double* U = F + (long)ncs * ld + ncs;
#pragma unroll
    for (int t = 0; t < 4; t++)
    {
        const int i = 64 * ti + 16 * t + ln;
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int j = 64 * tj + 16 * w + lk + 4 * q;
            if (i < nt && j < nrs && i >= j)
                U[(long)j * ld + i] -= acc[t][q];
        }
    }
    __syncthreads(); // panels are reused by the next tile of this workgroup
}

// backward substitution of one front: x_J = L11^-T (y_J - L21^T x_R)
__device__ void dev_backward(const CholPlanDev& p, const double* __restrict__ fronts, int f,
                             double* __restrict__ lds, double* __restrict__ xnew,
                             double* __restrict__ xout)
{
    const int ncb = p.ncb[f], nb = p.nb[f];
    const long ld = 6L * nb + 1;
    const int ncs = 6 * ncb, nrs = 6 * (nb - ncb);
    const double* F = fronts + p.off[f];
    const int ldsl = ncs + 1;
    double* Ls = lds;                  // ncs*(ncs+1)
    double* vs = lds + ncs * ldsl;     // ncs
    double* xr = vs + ncs;             // nrs
    const int32_t* rows = p.rows + p.rows_ptr[f];
    dev_load_l11(F, ld, ncs, Ls);
    for (int i = threadIdx.x; i < nrs; i += CBS)
        xr[i] = xnew[6L * rows[i / 6] + (i % 6)];
    __syncthreads();
    { // v_j = y_j - sum_i L21[i,j] x_R[i]: one wave per column, lanes stride the rows
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int j = w; j < ncs; j += CBS / 64)
        {
            const double* col = F + (long)j * ld + ncs;
            double s = 0;
            for (int i = lane; i < nrs; i += 64)
                s += col[i] * xr[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                s += __shfl_down(s, off, 64);
            if (lane == 0)
                vs[j] = F[(long)j * ld + (ld - 1)] - s;
        }
    }
    __syncthreads();
    for (int j0 = ncs - 6; j0 >= 0; j0 -= 6)
    { // L11^T x = v, 6 unknowns at a time (upper-triangular 6x6 solve by one thread)
        if (threadIdx.x == 0)
        {
            double l[6][6], v[6];
#pragma unroll
            for (int j = 0; j < 6; j++)
            {
                v[j] = vs[j0 + j];
#pragma unroll
                for (int k = 0; k < 6; k++)
                    l[k][j] = (k >= j) ? Ls[(j0 + j) * ldsl + j0 + k] : 0.0; // L[row k][col j]
            }
#pragma unroll
            for (int j = 5; j >= 0; j--)
            {
                double s = v[j];
#pragma unroll
                for (int k = 0; k < 6; k++)
                    if (k > j)
                        s -= l[k][j] * v[k];
                v[j] = s / l[j][j];
            }
#pragma unroll
            for (int j = 0; j < 6; j++)
                vs[j0 + j] = v[j];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < j0; t += CBS)
        {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 6; k++)
                s += Ls[t * ldsl + j0 + k] * vs[j0 + k];
            vs[t] -= s;
        }
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

// ---------------------------------------------------------------- stage 0: subtrees ----
__global__ __launch_bounds__(CBS) void k_subtree_factor(CholPlanDev p, double* __restrict__ fronts,
                                                        int task0, int32_t* __restrict__ fail)
{
    extern __shared__ double lds[];
    const int task = task0 + blockIdx.x;
    for (int fi = p.task_ptr[task]; fi < p.task_ptr[task + 1]; fi++)
    {
        const int f = p.task_fronts[fi];
        const int ncb = p.ncb[f], nb = p.nb[f];
        const long ld = 6L * nb + 1;
        const int ncs = 6 * ncb, nrs = 6 * (nb - ncb), nt = nrs + 1;
        double* F = fronts + p.off[f];
        double* Ls = lds;
        double* Bt = lds + ncs * (ncs + 1);
        dev_extend_add(p, fronts, f, 0, nb);
        dev_potrf(F, ld, ncs, Ls, fail);
        __syncthreads();
        for (int r0 = 0; r0 < nt; r0 += TR)
            dev_trsm_tile(F, ld, ncs, ncs + r0, min(TR, nt - r0), Ls, Bt);
        __threadfence_block();
        __syncthreads();
        const int nti = (nt + 63) / 64, ntj = (nrs + 63) / 64;
        for (int tj = 0; tj < ntj; tj++)
            for (int ti = tj; ti < nti; ti++)
                dev_syrk_tile(F, ld, ncs, nt, nrs, ti, tj, lds);
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------- upper stages ---------
__global__ __launch_bounds__(CBS) void k_up_extend_add(CholPlanDev p, double* __restrict__ fronts,
                                                       const int32_t* __restrict__ wl)
{
    const int32_t* it = wl + 3 * blockIdx.x;
    dev_extend_add(p, fronts, it[0], it[1], it[2]);
}

__global__ __launch_bounds__(CBS) void k_up_potrf(CholPlanDev p, double* __restrict__ fronts,
                                                  int task0, int32_t* __restrict__ fail)
{
    extern __shared__ double lds[];
    const int f = p.task_fronts[p.task_ptr[task0 + blockIdx.x]];
    dev_potrf(fronts + p.off[f], 6L * p.nb[f] + 1, 6 * p.ncb[f], lds, fail);
}

__global__ __launch_bounds__(CBS) void k_up_trsm(CholPlanDev p, double* __restrict__ fronts,
                                                 const int32_t* __restrict__ wl)
{
    extern __shared__ double lds[];
    const int32_t* it = wl + 3 * blockIdx.x;
    const int f = it[0];
    const int ncs = 6 * p.ncb[f];
    const long ld = 6L * p.nb[f] + 1;
    double* F = fronts + p.off[f];
    double* Ls = lds;
    double* Bt = lds + ncs * (ncs + 1);
    dev_load_l11(F, ld, ncs, Ls);
    __syncthreads();
    dev_trsm_tile(F, ld, ncs, ncs + it[1], it[2] - it[1], Ls, Bt);
}

__global__ __launch_bounds__(CBS) void k_up_syrk(CholPlanDev p, double* __restrict__ fronts,
                                                 const int32_t* __restrict__ wl)
{
    extern __shared__ double lds[];
    const int32_t* it = wl + 3 * blockIdx.x;
    const int f = it[0];
    const int ncs = 6 * p.ncb[f], nrs = 6 * (p.nb[f] - p.ncb[f]);
    dev_syrk_tile(fronts + p.off[f], 6L * p.nb[f] + 1, ncs, nrs + 1, nrs, it[1], it[2], lds);
}

__global__ __launch_bounds__(CBS) void k_backward_stage(CholPlanDev p,
                                                        const double* __restrict__ fronts,
                                                        int task0, double* __restrict__ xnew,
                                                        double* __restrict__ xout)
{
    extern __shared__ double lds[];
    const int task = task0 + blockIdx.x;
    for (int fi = p.task_ptr[task + 1] - 1; fi >= p.task_ptr[task]; fi--)
        dev_backward(p, fronts, p.task_fronts[fi], lds, xnew, xout);
}

void ensure_lds(const void* fn, size_t bytes)
{
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

} // namespace

namespace cugo_k
{

size_t chol_lds_factor_bytes(int nc_max)
{
    const size_t trsm = (size_t)nc_max * (nc_max + 1) + (size_t)TR * (nc_max + 1);
    const size_t syrk = 2 * (size_t)nc_max * PST;
    return (std::max(trsm, syrk) + 8) * sizeof(double);
}
size_t chol_lds_backward_bytes(int nc_max, long ld_max)
{
    return (size_t)(nc_max * (nc_max + 1) + nc_max + ld_max + 8) * sizeof(double);
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

void launch_chol_subtree_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                               int ntasks, size_t lds_bytes, int32_t* d_fail)
{
    if (ntasks <= 0)
        return;
    ensure_lds(reinterpret_cast<const void*>(k_subtree_factor), lds_bytes);
    hipLaunchKernelGGL(k_subtree_factor, dim3(ntasks), dim3(CBS), lds_bytes, s, p, d_fronts, task0,
                       d_fail);
}

void launch_chol_upper_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                             int ntasks, const int32_t* d_wl, int ea0, int nea, int tr0, int ntr,
                             int sy0, int nsy, size_t lds_bytes, int32_t* d_fail)
{
    if (ntasks <= 0)
        return;
    if (nea > 0)
        hipLaunchKernelGGL(k_up_extend_add, dim3(nea), dim3(CBS), 0, s, p, d_fronts, d_wl + 3L * ea0);
    ensure_lds(reinterpret_cast<const void*>(k_up_potrf), lds_bytes);
    hipLaunchKernelGGL(k_up_potrf, dim3(ntasks), dim3(CBS), lds_bytes, s, p, d_fronts, task0, d_fail);
    if (ntr > 0)
    {
        ensure_lds(reinterpret_cast<const void*>(k_up_trsm), lds_bytes);
        hipLaunchKernelGGL(k_up_trsm, dim3(ntr), dim3(CBS), lds_bytes, s, p, d_fronts,
                           d_wl + 3L * tr0);
    }
    if (nsy > 0)
    {
        ensure_lds(reinterpret_cast<const void*>(k_up_syrk), lds_bytes);
        hipLaunchKernelGGL(k_up_syrk, dim3(nsy), dim3(CBS), lds_bytes, s, p, d_fronts,
                           d_wl + 3L * sy0);
    }
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
