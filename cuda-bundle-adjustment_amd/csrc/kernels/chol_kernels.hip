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
// Every building block is written for the latency regime these fronts live in (a few hundred
// rows, one or a handful of workgroups): no integer division in inner loops, independent
// loads issued together before their uses, sequential 6x6 solves in registers, 1024-thread
// workgroups (4 waves per SIMD) wherever a single workgroup sits on the critical path.

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// children -> parent, restricted to parent block columns [cb0, cb1); children one after the
// other (barrier in between) so every parent entry is summed in child order.  One wave per
// child column; each lane gathers up to 4 independent parent entries before storing them.
__device__ void dev_extend_add(const CholPlanDev& p, double* __restrict__ fronts, int f, int cb0,
                               int cb1)
{
    const long ldp = 6L * p.nb[f] + 1;
    double* Fp = fronts + p.off[f];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    for (int ci = p.child_ptr[f]; ci < p.child_ptr[f + 1]; ci++)
    {
        const int c = p.child[ci];
        const int ncb = p.ncb[c], nbr = p.nb[c] - ncb;
        const int32_t* rel = p.rel + p.rel_ptr[c];
        // child update block columns whose parent column falls in [cb0, cb1) (rel is ascending):
        // counted with wave ballots, one pass of independent loads
        int jlo = 0, jhi = 0;
        for (int base = 0; base < nbr; base += 64)
        {
            const int i = base + lane;
            const int rv = i < nbr ? rel[i] : 0x7fffffff;
            jlo += __popcll(__ballot(rv < cb0));
            jhi += __popcll(__ballot(rv < cb1));
        }
        if (jhi > jlo)
        {
            const long ldc = 6L * p.nb[c] + 1;
            const double* U = fronts + p.off[c] + (6L * ncb) * ldc + 6L * ncb; // (0,0) of update
            const int nru = 6 * nbr + 1;
            for (int jb = jlo; jb < jhi; jb++)
            {
                const long pjb = 6L * rel[jb];
                for (int jj = wv; jj < 6; jj += nwv)
                {
                    const int j = 6 * jb + jj;
                    const double* ucol = U + (long)j * ldc;
                    double* pcol = Fp + (pjb + jj) * ldp;
                    for (int i0 = j + lane; i0 < nru; i0 += 256)
                    {
                        long pi[4];
                        double u[4], v[4];
#pragma unroll
                        for (int t = 0; t < 4; t++)
                        {
                            const int i = i0 + 64 * t;
                            const bool ok = i < nru;
                            const int ib = ok ? i / 6 : 0;
                            pi[t] = !ok ? -1 : (i == nru - 1 ? ldp - 1 : 6L * rel[ib] + (i - 6 * ib));
                            u[t] = ok ? ucol[i] : 0.0;
                        }
#pragma unroll
                        for (int t = 0; t < 4; t++)
                            v[t] = pi[t] >= 0 ? pcol[pi[t]] : 0.0;
#pragma unroll
                        for (int t = 0; t < 4; t++)
                            if (pi[t] >= 0)
                                pcol[pi[t]] = v[t] + u[t];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// 1/sqrt(d) and sqrt(d) without the IEEE division / sqrt sequences (each ~150-250 cycles of
// dependent fp64 work on the critical path): v_rsq_f64 seed + two Newton steps, then one
// correction of the root.  Result within ~1 ulp.
__device__ __forceinline__ void rsqrt_sqrt(double d, double& rs, double& s)
{
    double r = __builtin_amdgcn_rsq(d);
    r = r * fma(-0.5 * d * r, r, 1.5);
    r = r * fma(-0.5 * d * r, r, 1.5);
    double q = d * r;
    q = fma(0.5 * r, fma(-q, q, d), q);
    rs = r;
    s = q;
}

// in-register Cholesky of the 6x6 diagonal block at (j0,j0) of an LDS matrix (one thread);
// also stores the reciprocals of the new diagonal entries in dinv[j0..j0+5]
__device__ __forceinline__ bool chol6_lds(double* __restrict__ Ls, int lds, int j0,
                                          double* __restrict__ dinv)
{
    double a[6][6];
#pragma unroll
    for (int c = 0; c < 6; c++)
#pragma unroll
        for (int r = 0; r < 6; r++)
            a[r][c] = (r >= c) ? Ls[(j0 + c) * lds + j0 + r] : 0.0;
    bool bad = false;
    double iv[6];
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
        double inv, sq;
        rsqrt_sqrt(d, inv, sq);
        a[j][j] = sq;
        iv[j] = inv;
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
    {
        dinv[j0 + c] = iv[c];
#pragma unroll
        for (int r = 0; r < 6; r++)
            if (r >= c)
                Ls[(j0 + c) * lds + j0 + r] = a[r][c];
    }
    return bad;
}

// lower triangle of F11 -> LDS (upper part zero); 32 lanes walk a column
__device__ void dev_load_l11(const double* __restrict__ F, long ld, int nc, double* __restrict__ Ls)
{
    const int lds = nc + 1;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, nty = blockDim.x >> 5;
    for (int c = ty; c < nc; c += nty)
        for (int r = tx; r < nc; r += 32)
            Ls[c * lds + r] = (r >= c) ? F[(long)c * ld + r] : 0.0;
}

// dinv[j] = 1 / L11[j][j] from an LDS copy of L11 (one division per thread, in parallel)
__device__ void dev_recip_diag(const double* __restrict__ Ls, int nc, double* __restrict__ dinv)
{
    for (int j = threadIdx.x; j < nc; j += blockDim.x)
        dinv[j] = 1.0 / Ls[j * (nc + 1) + j];
}

// L11 = chol(F11) in LDS (Ls: nc x nc, leading dimension nc+1), blocked by 6 columns.
// On return Ls holds L11 (lower) and F11 in global memory is overwritten with it.
__device__ void dev_potrf(double* __restrict__ F, long ld, int nc, double* __restrict__ Ls,
                          double* __restrict__ dinv, int32_t* __restrict__ fail)
{
    const int lds = nc + 1;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, nty = blockDim.x >> 5;
    dev_load_l11(F, ld, nc, Ls);
    __syncthreads();
    for (int j0 = 0; j0 < nc; j0 += 6)
    {
        if (threadIdx.x == 0)
        {
            if (chol6_lds(Ls, lds, j0, dinv))
                *fail = 1;
        }
        __syncthreads();
        const int m = nc - (j0 + 6); // rows below the diagonal block inside F11
        if ((int)threadIdx.x < m)
        { // panel rows: x L_D^T = a
            const int row = j0 + 6 + threadIdx.x;
            double x[6], a[6], dl[6][6], di[6];
#pragma unroll
            for (int c = 0; c < 6; c++)
            {
                a[c] = Ls[(j0 + c) * lds + row];
                di[c] = dinv[j0 + c];
#pragma unroll
                for (int k = 0; k < 6; k++)
                    dl[k][c] = (k < c) ? Ls[(j0 + k) * lds + j0 + c] : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 6; c++)
            {
                double s = a[c];
#pragma unroll
                for (int k = 0; k < 6; k++)
                    if (k < c)
                        s -= x[k] * dl[k][c];
                x[c] = s * di[c];
            }
#pragma unroll
            for (int c = 0; c < 6; c++)
                Ls[(j0 + c) * lds + row] = x[c];
        }
        __syncthreads();
        for (int c = j0 + 6 + ty; c < nc; c += nty)
        { // trailing update inside F11 (lower part): 32 lanes walk the rows of column c
            double pc[6];
#pragma unroll
            for (int k = 0; k < 6; k++)
                pc[k] = Ls[(j0 + k) * lds + c];
            for (int r = c + tx; r < nc; r += 32)
            {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 6; k++)
                    s += Ls[(j0 + k) * lds + r] * pc[k];
                Ls[c * lds + r] -= s;
            }
        }
        __syncthreads();
    }
    for (int c = ty; c < nc; c += nty)
        for (int r = c + tx; r < nc; r += 32)
            F[(long)c * ld + r] = Ls[c * lds + r];
}

// rows [row0, row0+nrows) (absolute scalar rows, nrows <= TR) of F21: X L11^T = B.
// Ls holds L11; Bt is a TR x (nc+1) LDS tile.  LPR adjacent lanes share one row, so a row
// never leaves its wave: LDS operations of one wave execute in order and a wave-level fence
// is all the synchronisation the column loop needs.  LPR = blockDim / 64 (4 or 16).
__device__ void dev_trsm_tile(double* __restrict__ F, long ld, int nc, long row0, int nrows,
                              const double* __restrict__ Ls, const double* __restrict__ dinv,
                              double* __restrict__ Bt)
{
    const int lds = nc + 1;
    {
        const int tr = threadIdx.x & 63, tc = threadIdx.x >> 6, ntc = blockDim.x >> 6;
        if (tr < nrows)
            for (int c = tc; c < nc; c += ntc)
                Bt[tr * lds + c] = F[(long)c * ld + row0 + tr];
    }
    __syncthreads();
    const int lpr = blockDim.x >> 6;                    // lanes per row
    const int shift = lpr == 16 ? 4 : 2;
    const int r = threadIdx.x >> shift, g = threadIdx.x & (lpr - 1);
    if (r < nrows)
    {
        double* row = Bt + r * lds;
        for (int j0 = 0; j0 < nc; j0 += 6)
        {
            // every lane of the row solves the 6x6 system redundantly (same inputs, same result)
            double x[6], a[6], dl[6][6], di[6];
#pragma unroll
            for (int c = 0; c < 6; c++)
            {
                a[c] = row[j0 + c];
                di[c] = dinv[j0 + c];
#pragma unroll
                for (int k = 0; k < 6; k++)
                    dl[k][c] = (k < c) ? Ls[(j0 + k) * lds + j0 + c] : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 6; c++)
            {
                double s = a[c];
#pragma unroll
                for (int k = 0; k < 6; k++)
                    if (k < c)
                        s -= x[k] * dl[k][c];
                x[c] = s * di[c];
            }
            wave_lds_sync(); // all lanes of the row have read the old values
            if (g == 0)
            {
#pragma unroll
                for (int c = 0; c < 6; c++)
                    row[j0 + c] = x[c];
            }
            for (int c = j0 + 6 + g; c < nc; c += lpr)
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
    {
        const int tr = threadIdx.x & 63, tc = threadIdx.x >> 6, ntc = blockDim.x >> 6;
        if (tr < nrows)
            for (int c = tc; c < nc; c += ntc)
                F[(long)c * ld + row0 + tr] = Bt[tr * lds + c];
    }
    __syncthreads();
}

// U(ti,tj) -= L21(ti rows) L21(tj rows)^T, 64x64 tiles on the f64 matrix cores, executed by
// TEAMS of 4 waves (a 256-thread workgroup is one team, a 1024-thread one four teams working
// on four tiles at once).  The team stages its two 64-row panels through LDS in K chunks of
// 24 (k-major, stride 80 doubles: the two 16-lane halves of a ds_read_b64 then hit different
// banks); all loads of a chunk are issued together.  tiles[] enumerates (ti,tj) with ti>=tj.
// MFMA operand map (v_mfma_f64_16x16x4_f64): lane l supplies A[m = l&15][k = l>>4] and
// B[k = l>>4][n = l&15]; result reg q holds D[m = (l>>4) + 4q][n = l&15].  With m = U column
// and n = U row the 16 lanes l&15 hit consecutive rows of one column: 128-B segments.
constexpr int PST = 80; // LDS panel stride (doubles) per k
constexpr int KC = 24;  // K chunk
constexpr int TEAM_LDS = 2 * KC * PST; // doubles per team

__device__ void dev_syrk_tiles(double* __restrict__ F, long ld, int ncs, int nt, int nrs,
                               int first_tile, int ntiles_total, int ntj, double* __restrict__ lds)
{
    // tile index -> (ti, tj): column-major over the lower triangle of tiles, nti = ceil(nt/64)
    const int nti = (nt + 63) >> 6;
    const int team = threadIdx.x >> 8, nteams = blockDim.x >> 8;
    const int tt = threadIdx.x & 255;
    const int lane = tt & 63, w = tt >> 6;
    const int ln = lane & 15, lk = lane >> 4;
    const double* L21 = F + ncs; // element (row i, col k) = L21[k*ld + i]
    double* Pi = lds + team * TEAM_LDS;
    double* Pj = Pi + KC * PST;
    double* U = F + (long)ncs * ld + ncs;
    for (int base = first_tile; base < ntiles_total; base += nteams)
    {
        const int tile = base + team;
        const bool active = tile < ntiles_total;
        int ti = 0, tj = 0;
        if (active)
        { // unrank: columns tj = 0.. have (nti - tj) tiles each
            int rem = tile;
            while (tj < ntj && rem >= nti - tj)
            {
                rem -= nti - tj;
                tj++;
            }
            ti = tj + rem;
        }
        const bool diag = ti == tj;
        double4_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
            acc[t] = double4_t{0, 0, 0, 0};
        // prefetch the U entries this lane will update (independent loads, issued first)
        double uold[4][4];
        if (active)
        {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    const int i = 64 * ti + 16 * t + ln, j = 64 * tj + 16 * w + lk + 4 * q;
                    uold[t][q] = (i < nt && j < nrs && i >= j) ? U[(long)j * ld + i] : 0.0;
                }
        }
        for (int kc = 0; kc < ncs; kc += KC)
        {
            const int kn = min(KC, ncs - kc);
            if (active)
            {
                const int r = tt & 63;
                const int gi = 64 * ti + r, gj = 64 * tj + r;
                for (int k = tt >> 6; k < kn; k += 4)
                {
                    Pi[k * PST + r] = gi < nt ? L21[(long)(kc + k) * ld + gi] : 0.0;
                    if (!diag)
                        Pj[k * PST + r] = gj < nt ? L21[(long)(kc + k) * ld + gj] : 0.0;
                }
            }
            __syncthreads();
            if (active)
            {
                const double* Pa = diag ? Pi : Pj;
                for (int k0 = 0; k0 < kn; k0 += 4)
                {
                    const int k = k0 + lk;
                    const bool kok = k < kn;
                    const double a = kok ? Pa[k * PST + 16 * w + ln] : 0.0;
#pragma unroll
                    for (int t = 0; t < 4; t++)
                    {
                        const double b = kok ? Pi[k * PST + 16 * t + ln] : 0.0;
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
        if (active)
        {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    const int i = 64 * ti + 16 * t + ln, j = 64 * tj + 16 * w + lk + 4 * q;
                    if (i < nt && j < nrs && i >= j)
                        U[(long)j * ld + i] = uold[t][q] - acc[t][q];
                }
        }
    }
}

// value of a double held by lane `src` (wave-uniform, in an SGPR): two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int src)
{
    const int s = __builtin_amdgcn_readfirstlane(src);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), s);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), s);
    return __hiloint2double(hi, lo);
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
    double* dinv = vs + ncs;           // ncs
    double* xr = dinv + ncs;           // nrs
    const int32_t* rows = p.rows + p.rows_ptr[f];
    dev_load_l11(F, ld, ncs, Ls);
    for (int i = threadIdx.x; i < nrs; i += blockDim.x)
    {
        const int ib = i / 6;
        xr[i] = xnew[6L * rows[ib] + (i - 6 * ib)];
    }
    __syncthreads();
    dev_recip_diag(Ls, ncs, dinv);
    { // v_j = y_j - sum_i L21[i,j] x_R[i]: 16 lanes per column, lanes stride the rows
        const int g = threadIdx.x >> 4, l16 = threadIdx.x & 15, ng = blockDim.x >> 4;
        for (int j = g; j < ncs; j += ng)
        {
            const double* col = F + (long)j * ld + ncs;
            double s = 0;
            for (int i = l16; i < nrs; i += 64)
            { // four independent loads in flight per lane
                const double a0 = col[i];
                const double a1 = i + 16 < nrs ? col[i + 16] : 0.0;
                const double a2 = i + 32 < nrs ? col[i + 32] : 0.0;
                const double a3 = i + 48 < nrs ? col[i + 48] : 0.0;
                s += a0 * xr[i];
                if (i + 16 < nrs)
                    s += a1 * xr[i + 16];
                if (i + 32 < nrs)
                    s += a2 * xr[i + 32];
                if (i + 48 < nrs)
                    s += a3 * xr[i + 48];
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1)
                s += __shfl_down(s, off, 16);
            if (l16 == 0)
                vs[j] = F[(long)j * ld + (ld - 1)] - s;
        }
    }
    __syncthreads();
    // L11^T x = v by ONE wave, wave-synchronously: lane t keeps v[t] and v[t+64] in registers,
    // x_j is broadcast with a lane read, the row of L for the next step is prefetched from LDS
    // (stride ncs+1 doubles: conflict-free).  No workgroup barrier inside the 6*ncb steps.
    if (threadIdx.x < 64)
    {
        const int lane = threadIdx.x;
        double v0 = lane < ncs ? vs[lane] : 0.0;
        double v1 = lane + 64 < ncs ? vs[lane + 64] : 0.0;
        int j = ncs - 1;
        double l0 = lane < j ? Ls[lane * ldsl + j] : 0.0;
        double l1 = lane + 64 < j ? Ls[(lane + 64) * ldsl + j] : 0.0;
        double dj = dinv[j];
        for (; j >= 0; j--)
        {
            // prefetch row j-1
            const int jn = j - 1;
            double n0 = 0.0, n1 = 0.0, dn = 0.0;
            if (jn >= 0)
            {
                n0 = lane < jn ? Ls[lane * ldsl + jn] : 0.0;
                n1 = lane + 64 < jn ? Ls[(lane + 64) * ldsl + jn] : 0.0;
                dn = dinv[jn];
            }
            const double vj = j < 64 ? readlane_f64(v0, j) : readlane_f64(v1, j - 64);
            const double xj = vj * dj;
            v0 -= l0 * xj;
            v1 -= l1 * xj;
            if (lane == (j & 63))
            {
                if (j < 64)
                    v0 = xj;
                else
                    v1 = xj;
            }
            l0 = n0, l1 = n1, dj = dn;
        }
        if (lane < ncs)
            vs[lane] = v0;
        if (lane + 64 < ncs)
            vs[lane + 64] = v1;
    }
    __syncthreads();
    const int c0 = p.col0[f];
    for (int j = threadIdx.x; j < ncs; j += blockDim.x)
    {
        const int jb = j / 6, comp = j - 6 * jb;
        const double v = vs[j];
        xnew[6L * (c0 + jb) + comp] = v;
        xout[6L * p.perm[c0 + jb] + comp] = v;
    }
    __threadfence_block();
    __syncthreads();
}

constexpr int BIG = 1024; // workgroup size of the latency-critical single-front kernels

// ---------------------------------------------------------------- stage 0: subtrees ----
__global__ __launch_bounds__(BIG) void k_subtree_factor(CholPlanDev p, double* __restrict__ fronts,
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
        double* dinv = lds + ncs * (ncs + 1);
        double* Bt = dinv + ncs;
        dev_extend_add(p, fronts, f, 0, nb);
        dev_potrf(F, ld, ncs, Ls, dinv, fail);
        __syncthreads();
        for (int r0 = 0; r0 < nt; r0 += TR)
            dev_trsm_tile(F, ld, ncs, ncs + r0, min(TR, nt - r0), Ls, dinv, Bt);
        __threadfence_block();
        __syncthreads();
        const int nti = (nt + 63) / 64, ntj = (nrs + 63) / 64;
        int ntiles = 0;
        for (int tj = 0; tj < ntj; tj++)
            ntiles += nti - tj;
        dev_syrk_tiles(F, ld, ncs, nt, nrs, 0, ntiles, ntj, lds);
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

__global__ __launch_bounds__(BIG) void k_up_potrf(CholPlanDev p, double* __restrict__ fronts,
                                                  int task0, int32_t* __restrict__ fail)
{
    extern __shared__ double lds[];
    const int f = p.task_fronts[p.task_ptr[task0 + blockIdx.x]];
    const int ncs = 6 * p.ncb[f];
    dev_potrf(fronts + p.off[f], 6L * p.nb[f] + 1, ncs, lds, lds + ncs * (ncs + 1), fail);
}

// trsm tiles (touch the pivot columns) and, in the same launch, the extend-add of the
// boundary columns (touch the update region): independent data, one kernel boundary less
__global__ __launch_bounds__(BIG) void k_up_trsm(CholPlanDev p, double* __restrict__ fronts,
                                                 const int32_t* __restrict__ wl, int ntr,
                                                 const int32_t* __restrict__ wl_ea)
{
    extern __shared__ double lds[];
    if ((int)blockIdx.x >= ntr)
    {
        const int32_t* it = wl_ea + 3 * (blockIdx.x - ntr);
        dev_extend_add(p, fronts, it[0], it[1], it[2]);
        return;
    }
    const int32_t* it = wl + 3 * blockIdx.x;
    const int f = it[0];
    const int ncs = 6 * p.ncb[f];
    const long ld = 6L * p.nb[f] + 1;
    double* F = fronts + p.off[f];
    double* Ls = lds;
    double* dinv = lds + ncs * (ncs + 1);
    double* Bt = dinv + ncs;
    dev_load_l11(F, ld, ncs, Ls);
    __syncthreads();
    dev_recip_diag(Ls, ncs, dinv);
    __syncthreads();
    dev_trsm_tile(F, ld, ncs, ncs + it[1], it[2] - it[1], Ls, dinv, Bt);
}

// one workgroup (one team) per 64x64 tile; it[1] = linear tile index
__global__ __launch_bounds__(CBS) void k_up_syrk(CholPlanDev p, double* __restrict__ fronts,
                                                 const int32_t* __restrict__ wl)
{
    extern __shared__ double lds[];
    const int32_t* it = wl + 3 * blockIdx.x;
    const int f = it[0];
    const int ncs = 6 * p.ncb[f], nrs = 6 * (p.nb[f] - p.ncb[f]);
    dev_syrk_tiles(fronts + p.off[f], 6L * p.nb[f] + 1, ncs, nrs + 1, nrs, it[1], it[1] + 1,
                   (nrs + 63) / 64, lds);
}

__global__ __launch_bounds__(BIG) void k_backward_stage(CholPlanDev p,
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
    const size_t trsm = (size_t)nc_max * (nc_max + 2) + (size_t)TR * (nc_max + 1);
    const size_t syrk = 4 * (size_t)TEAM_LDS; // four teams in the 1024-thread subtree kernel
    return (std::max(trsm, syrk) + 8) * sizeof(double);
}
size_t chol_lds_backward_bytes(int nc_max, long ld_max)
{
    return (size_t)(nc_max * (nc_max + 1) + 2 * nc_max + ld_max + 8) * sizeof(double);
}

void launch_chol_assemble(hipStream_t s, const CholPlanDev& p, double* d_fronts,
                          size_t front_doubles, const double* d_Hsc, double lambda,
                          const double* d_bsc)
{
    (void)hipMemsetAsync(d_fronts, 0, front_doubles * sizeof(double), s);
    const long n = 36L * p.n_hsc_blocks;
    if (n > 0)
        CUGO_LAUNCH(k_assemble_blocks, dim3((unsigned)((n + CBS - 1) / CBS)), dim3(CBS), 0, s,
                           p, d_fronts, d_Hsc, lambda);
    if (p.n > 0)
        CUGO_LAUNCH(k_assemble_rhs, dim3((6 * p.n + CBS - 1) / CBS), dim3(CBS), 0, s, p,
                           d_fronts, d_bsc);
}

void launch_chol_subtree_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                               int ntasks, size_t lds_bytes, int32_t* d_fail)
{
    if (ntasks <= 0)
        return;
    ensure_lds(reinterpret_cast<const void*>(k_subtree_factor), lds_bytes);
    CUGO_LAUNCH(k_subtree_factor, dim3(ntasks), dim3(BIG), lds_bytes, s, p, d_fronts, task0,
                       d_fail);
}

void launch_chol_upper_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                             int ntasks, const int32_t* d_wl, int eap0, int neap, int ea0, int nea,
                             int tr0, int ntr, int sy0, int nsy, size_t lds_bytes, int32_t* d_fail)
{
    if (ntasks <= 0)
        return;
    if (neap > 0) // children -> pivot columns, one workgroup per 2 block columns
        CUGO_LAUNCH(k_up_extend_add, dim3(neap), dim3(CBS), 0, s, p, d_fronts,
                           d_wl + 3L * eap0);
    ensure_lds(reinterpret_cast<const void*>(k_up_potrf), lds_bytes);
    CUGO_LAUNCH(k_up_potrf, dim3(ntasks), dim3(BIG), lds_bytes, s, p, d_fronts, task0, d_fail);
    if (ntr + nea > 0)
    {
        ensure_lds(reinterpret_cast<const void*>(k_up_trsm), lds_bytes);
        CUGO_LAUNCH(k_up_trsm, dim3(ntr + nea), dim3(BIG), lds_bytes, s, p, d_fronts,
                           d_wl + 3L * tr0, ntr, d_wl + 3L * ea0);
    }
    if (nsy > 0)
    {
        CUGO_LAUNCH(k_up_syrk, dim3(nsy), dim3(CBS), TEAM_LDS * sizeof(double), s, p, d_fronts,
                           d_wl + 3L * sy0);
    }
}

void launch_chol_backward_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                                int ntasks, size_t lds_bytes, double* d_xnew, double* d_x)
{
    if (ntasks <= 0)
        return;
    ensure_lds(reinterpret_cast<const void*>(k_backward_stage), lds_bytes);
    CUGO_LAUNCH(k_backward_stage, dim3(ntasks), dim3(BIG), lds_bytes, s, p, d_fronts, task0,
                       d_xnew, d_x);
}

} // namespace cugo_k
