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
//                         (off by default, CUGO_MIN_SUBTREE_TASKS=0 enables it; k_subtree_factor)
//   upper stages        : one etree level per stage, two batched kernels per level so that a
//                         big front is spread over many workgroups (256 CUs / 8 XCDs):
//        k_up_potrf       1 workgroup / front: extend-add of the children into F11, F11 = L D L^T in LDS
//                         by 16-column register panels, W = L11^-1 = D^-1/2 L^-1 built behind the panels
//                         (dev_potrf16; CUGO_PANEL16=0: 6-column L L^T panels, then W in a phase of its
//                         own); extra workgroups of the same launch do the extend-add of everything below F11
//        k_up_trsyrk      per 64x64 tile of the update matrix: X = B W^T for its two L21 row
//                         tiles, U -= X_i X_j^T, all v_mfma_f64_16x16x4_f64 (k_up_trsyrk32: 32x32
//                         tiles on the levels with few fronts); the items of a front share an XCD
//        (k_up_potrf_la / k_up_lead: the opt-in look-ahead schedule, CUGO_LOOKAHEAD=1)
//   backward            : k_backward_stage per level, x_J = W^T (y_J - L21^T x_R): two mat-vecs;
//                         the ancestor part of L21^T x_R is done one launch ahead (extra workgroups)
//   before every factorisation: k_assemble_fronts — ONE launch that zeroes every lower-triangle entry no Hsc block
//                         or right-hand side entry lands on and scatters Hsc (+ lambda), the right-hand side and
//                         the reset of the zero-pivot flag (CUGO_ASM_FRONTS=0: k_clear_fronts + k_assemble_blocks)
//   (levels with more than 260 tiles of 64x64: k_up_trsm + k_up_syrk, every L21 row tile solved once)
//
// The right-hand side rides along as the last row of every front, so L y = b is a by-product
// of the factorisation (y ends in the rhs row of the pivot columns); only the backward
// substitution needs its own top-down pass.  All sums have a fixed order: bit-reproducible.
// A pivot <= 1e-14 (or NaN) raises *fail (ref: csrcholZeroPivot tol, src/cholesky.hpp:85).
#include <algorithm>
#include <cstdio>

#include "kernels.h"

namespace
{

constexpr int CBS = 256;
constexpr double PIVOT_TOL = 1e-14;
constexpr int TR = 64; // trsm / syrk tile edge
constexpr int NC_MAX = 96; // widest pivot block (scalars): L11 lives in LDS
// leading dimension of the LDS copy of L11: odd, and 16 rows longer than the matrix — a 16x16 tile of the
// trailing update that starts inside the matrix may hang over its lower edge into padding nobody reads,
// so only the one tile per panel that also hangs over the RIGHT edge needs per-element conditions
constexpr int LLD = NC_MAX + 17;
// Panel width of the in-LDS Cholesky (must divide 6).  The redundant in-register factorisation of
// the PW x PW diagonal block grows with PW^3 while the per-panel barrier / look-ahead / LDS
// round-trip overhead is paid 6/PW times per pose block: measured, 6 beats 3 (32 vs 38 us per
// 96-column front); 12 would not fit the 128-VGPR budget of a 1024-thread workgroup.
constexpr int PW = 6;

using cugo_k::CholPlanDev;
typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- assembly -------------
// workgroups [0, nblk): the Hsc blocks (+lambda on the diagonal); the ones after them: the right-hand
// side and the reset of the zero-pivot flag (one launch instead of two)
__device__ __forceinline__ void assemble_scatter(const CholPlanDev& p, double* __restrict__ fronts,
                                                 const double* __restrict__ Hsc, double lambda,
                                                 const double* __restrict__ bsc, int32_t* __restrict__ fail, int nblk,
                                                 int bid)
{
    if (bid >= nblk)
    {
        const int j = (bid - nblk) * CBS + threadIdx.x;
        if (j == 0)
            *fail = 0; // the zero-pivot flag of this factorisation
        if (j >= 6 * p.n)
            return;
        const int jb = j / 6, comp = j % 6;
        const int f = p.col_front[jb];
        const long ld = p.ldf[f];
        const long lc = 6L * (jb - p.col0[f]) + comp;
        fronts[p.off[f] + lc * ld + 6L * p.nb[f]] = bsc[6L * p.perm[jb] + comp]; // rhs row = row 6*nb
        return;
    }
    const long idx = (long)bid * CBS + threadIdx.x;
    if (idx >= 36L * p.n_hsc_blocks)
        return;
    const int k = (int)(idx / 36), t = (int)(idx % 36);
    const int r = t % 6, c = t / 6;
    const int f = p.blk_front[k];
    const long ld = p.ldf[f];
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

__global__ __launch_bounds__(CBS) void k_assemble_blocks(CholPlanDev p, double* __restrict__ fronts,
                                                         const double* __restrict__ Hsc,
                                                         double lambda, const double* __restrict__ bsc,
                                                         int32_t* __restrict__ fail, int nblk)
{
    assemble_scatter(p, fronts, Hsc, lambda, bsc, fail, nblk, (int)blockIdx.x);
}

// lower triangle (rows >= column) of 16 columns of one front := 0.  The strict upper triangles are
// cleared once, when the plan is uploaded, and never written afterwards (every global store of the
// factorisation is masked to row >= column), so this is the whole-buffer memset at half the bytes.
__global__ __launch_bounds__(CBS) void k_clear_fronts(CholPlanDev p, double* __restrict__ fronts,
                                                      const int32_t* __restrict__ items)
{
    const int32_t* it = items + 3 * blockIdx.x;
    const int f = it[0], c0 = it[1], c1 = it[2];
    const long ld = p.ldf[f];
    const int nrows = 6 * p.nb[f] + 1;
    double* F = fronts + p.off[f];
    for (int r = c0 + threadIdx.x; r < nrows; r += CBS)
#pragma unroll 4
        for (int c = c0; c < c1; c++)
            if (r >= c)
                F[(long)c * ld + r] = 0.0;
}

// Assembly that leaves nothing to clear, in ONE launch: the first `nzero` workgroups write zeros to every
// lower-triangle entry of the stored fronts that no Hsc block and no right-hand side entry lands on (two block
// columns of a front each; CholPlan::asm_map says which block positions are taken — most items hold none and
// need no map), the others scatter the Hsc blocks (+lambda) and the right-hand side as k_assemble_blocks does.
// The two kinds write disjoint entries, so no order between them is needed: every entry is written once
// instead of cleared by one launch and patched by the next.
// One 64-byte record per zero item (CholPlanDev::fat): front, block columns, "some position taken", block
// rows, map offset, storage offset, leading dimension.
__global__ __launch_bounds__(CBS) void k_assemble_fronts(CholPlanDev p, double* __restrict__ fronts,
                                                         const double* __restrict__ Hsc, double lambda,
                                                         const double* __restrict__ bsc, int32_t* __restrict__ fail,
                                                         int item0, int nzero, int nblk)
{
    if ((int)blockIdx.x >= nzero)
    {
        assemble_scatter(p, fronts, Hsc, lambda, bsc, fail, nblk, (int)blockIdx.x - nzero);
        return;
    }
    const int32_t* it = p.fat + 16L * (item0 + (int)blockIdx.x);
    const int4 h0 = *reinterpret_cast<const int4*>(it), h1 = *reinterpret_cast<const int4*>(it + 4);
    const long off = *reinterpret_cast<const int64_t*>(it + 8), ld = *reinterpret_cast<const int64_t*>(it + 10);
    const int cb0 = h0.y, cb1 = h0.z, nb = h1.y;
    const int nrows = 6 * nb + 1;
    double* F = fronts + off;
    if (!h1.x)
    { // no position of these columns is taken
        for (int r = 6 * cb0 + threadIdx.x; r < nrows; r += CBS)
            for (int c = 6 * cb0; c < 6 * cb1; c++)
                if (r >= c)
                    F[(long)c * ld + r] = 0.0;
        return;
    }
    const int32_t* map = p.asm_map + (long)(uint32_t)h1.z + ((long)h1.w << 32);
    for (int r = 6 * cb0 + threadIdx.x; r < nrows; r += CBS)
    {
        const int rb = r / 6, i = r - 6 * rb;
        for (int cb = cb0; cb < cb1; cb++)
        {
            if (rb < cb)
                continue;
            const int taken = rb == nb ? map[cb] : map[nb + cb * nb - cb * (cb - 1) / 2 + (rb - cb)];
            if (taken >= 0)
                continue; // the scatter writes this block / these rhs entries
            double* Fc = F + 6L * cb * ld + r;
#pragma unroll
            for (int c = 0; c < 6; c++)
                if (rb > cb || i >= c)
                    Fc[c * ld] = 0.0;
        }
    }
}

// Diagnostic phase stamps: only in a build with -DCUGO_STAMPS (make STAMPS=1) and run with
// CUGO_DEBUG_STAMPS=1.  Workgroup 0 of a kernel stores s_memtime at a few points into a side
// buffer that no other code reads.  In the normal build stamp() is an empty inline function
// (a run-time null check would cost a dependent global load per call).
#ifdef CUGO_STAMPS
__device__ long long* g_stamps = nullptr;
__device__ int g_stamp_block[8] = {0, 0, 0, 0, 1, 0, 0, 0}; // per kernel: the workgroup that stamps
__device__ __forceinline__ void stamp(int kernel, int slot)
{
    if (g_stamps && (int)blockIdx.x == g_stamp_block[kernel] && threadIdx.x == 0)
        g_stamps[kernel * 8 + slot] = clock64();
}
__device__ __forceinline__ void stamp_value(int kernel, int slot, long long v)
{
    if (g_stamps && (int)blockIdx.x == g_stamp_block[kernel] && threadIdx.x == 0)
        g_stamps[kernel * 8 + slot] = v;
}
#else
__device__ __forceinline__ void stamp(int, int) {}
__device__ __forceinline__ void stamp_value(int, int, long long) {}
#endif

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
// One 64-row chunk of one child update block column (6 scalar columns) added into the parent:
// lane <-> row, the 6 columns are 6 independent read-modify-writes in flight per lane.
__device__ __forceinline__ void ea_chunk(const double* __restrict__ U, long ldc, int nru, int nbr,
                                         const int32_t* __restrict__ rel, double* __restrict__ Fp,
                                         long ldp, long rhs_row, int jb, int ch, int lane,
                                         double* __restrict__ sink, double* dsink, int rbeg, int rend)
{
    // rows [rbeg, rend) of the child's update block column jb (rbeg >= 6*jb, rend <= nru)
    const int i = rbeg + 64 * ch + lane;
    const bool ok = i < rend;
    const int ic = ok ? i : 6 * jb;
    const int ib = ic / 6;
    const long pjb = 6L * rel[jb];
    const int rr = rel[min(ib, nbr - 1)];
    const long pi = (ic == nru - 1) ? rhs_row : 6L * rr + (ic - 6 * ib); // last row = rhs row
    double* dst[6];
    double u[6], v[6];
#pragma unroll
    for (int jj = 0; jj < 6; jj++)
    { // branch-free: masked lanes (past the end / above the diagonal) use their private sink slot
        const int j = 6 * jb + jj;
        const bool okj = ok && ic >= j;
        const double* src = okj ? U + (long)j * ldc + ic : sink;
        dst[jj] = okj ? Fp + (pjb + jj) * ldp + pi : dsink;
        u[jj] = *src;
    }
#pragma unroll
    for (int jj = 0; jj < 6; jj++)
        v[jj] = *dst[jj];
#pragma unroll
    for (int jj = 0; jj < 6; jj++)
        *dst[jj] = v[jj] + u[jj];
}

// Extend-add of the children of front f into its block columns [cb0, cb1).  Children are applied
// one after the other in list order (fixed summation order => bit-reproducible); within a child
// the (block column, 64-row chunk) units are dealt round-robin to the waves.
// `part`: 0 = all rows; 1 = only the child rows that map into the parent's PIVOT rows (the parent's
// F11: what its potrf needs); 2 = only the rows below them (F21 / F22).  Parts 1 and 2 touch
// disjoint parent entries, so they may run in different workgroups of one launch.
// The look-ahead schedule cuts part 2 once more at the parent's LEAD rows (its first la_np boundary
// block rows: the ones inside the pivot columns of ITS parent): 3 = the rows that map into the lead
// rows (what the lead block of the parent's update matrix needs), 4 = the rows below them.
constexpr int EA_BATCH = 32;
// dst_lds != nullptr (part 1 only): the contributions go into the LDS copy of F11 (leading dimension
// LLD) instead of the front in memory — the potrf workgroup then never writes F11 back and reads it
// again (a store -> load round trip through memory at the head of the critical kernel of the level).
template <bool TO_LDS = false>
__device__ __forceinline__ void dev_extend_add(const CholPlanDev& p, double* __restrict__ fronts, int f, int cb0,
                               int cb1, int part, double* dst_lds = nullptr, double* sink_lds = nullptr)
{
    __shared__ int s_child[EA_BATCH][5]; // child front, first / past-last matching update block column,
                                         // number of child rows (blocks) inside the parent's pivot block,
                                         // ... inside its pivot block or lead rows
    const int ncbp = p.ncb[f];
    const int nlead = ncbp + (part >= 3 ? p.la_np[f] : 0);
    // (a compile-time choice: with one destination pointer that may be either, every access would be
    // a flat one)
    const long ldp = TO_LDS ? (long)LLD : p.ldf[f], rhs_row = 6L * p.nb[f];
    double* Fp = TO_LDS ? dst_lds : fronts + p.off[f];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    double* sink = p.junk + ((blockIdx.x & 63) << 10) + threadIdx.x;
    const int c0 = p.child_ptr[f], c1 = p.child_ptr[f + 1];
    for (int cbase = c0; cbase < c1; cbase += EA_BATCH)
    {
        const int nchild = min(EA_BATCH, c1 - cbase);
        // child update block columns whose parent column falls in [cb0, cb1) (rel is ascending),
        // counted with wave ballots; one wave per child so the dependent metadata loads of
        // different children overlap
        for (int k = wv; k < nchild; k += nwv)
        {
            const int c = p.child[cbase + k];
            const int nbr = p.nb[c] - p.ncb[c];
            const int32_t* rel = p.rel + p.rel_ptr[c];
            int jlo = 0, jhi = 0, nsp = 0, nsl = 0;
            for (int base = 0; base < nbr; base += 64)
            {
                const int i = base + lane;
                const int rv = rel[min(i, nbr - 1)];
                jlo += __popcll(__ballot(i < nbr && rv < cb0));
                jhi += __popcll(__ballot(i < nbr && rv < cb1));
                nsp += __popcll(__ballot(i < nbr && rv < ncbp));
                nsl += __popcll(__ballot(i < nbr && rv < nlead));
            }
            if (lane == 0)
                s_child[k][0] = c, s_child[k][1] = jlo, s_child[k][2] = jhi, s_child[k][3] = nsp, s_child[k][4] = nsl;
        }
        __syncthreads();
        for (int k = 0; k < nchild; k++)
        {
            const int c = s_child[k][0], jlo = s_child[k][1], jhi = s_child[k][2];
            if (jhi <= jlo)
                continue; // uniform over the workgroup
            const int ncb = p.ncb[c], nbr = p.nb[c] - ncb;
            const long ldc = p.ldf[c];
            const double* U = fronts + p.off[c] + (6L * ncb) * ldc + 6L * ncb; // (0,0) of update
            const int32_t* rel = p.rel + p.rel_ptr[c];
            const int nru = 6 * nbr + 1;
            const int isplit = 6 * s_child[k][3]; // child rows below this index map into parent pivot rows
            const int isplit2 = 6 * s_child[k][4]; // ... into its pivot or lead rows
            int u = wv;
            for (int jb = jlo; jb < jhi; jb++)
            {
                const int rbeg = part == 4 ? max(6 * jb, isplit2) : part >= 2 ? max(6 * jb, isplit) : 6 * jb;
                const int rend = part == 1 ? min(nru, isplit) : part == 3 ? min(nru, isplit2) : nru;
                const int nch = rend > rbeg ? (rend - rbeg + 63) >> 6 : 0;
                for (; u < nch; u += nwv)
                    ea_chunk(U, ldc, nru, nbr, rel, Fp, ldp, rhs_row, jb, u, lane, sink, TO_LDS ? sink_lds : sink, rbeg,
                             rend);
                u -= nch;
            }
            __syncthreads(); // the next child may touch the same parent entries
        }
        __syncthreads(); // s_child is rewritten by the next batch
    }
}

// Part 1 into the LDS copy of F11, from the host-built child records CholPlanDev::ea1 (per child of
// the front, in child order: {child, boundary block rows, leading rows inside this front's pivots,
// offset of its rel list, offset of its update block (int64), its leading dimension (int64)}): what
// the general routine finds by walking child_ptr -> child -> five per-front arrays and counting over
// rel is known when the plan is built — three dependent round trips less before the first panel.
__device__ __forceinline__ void dev_extend_add_lead(const CholPlanDev& p, const double* __restrict__ fronts, int e0,
                                                    int e1, double* __restrict__ Ls, double* sink_lds)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    double* sink = p.junk + ((blockIdx.x & 63) << 10) + threadIdx.x;
    for (int k = e0; k < e1; k++)
    {
        const int32_t* d = p.ea1 + 8 * k;
        const long* d64 = reinterpret_cast<const long*>(d + 4);
        const int nbr = d[1], np = d[2];
        const int32_t* rel = p.rel + d[3];
        const double* U = fronts + d64[0];
        const long ldc = d64[1];
        const int nru = 6 * nbr + 1, isplit = 6 * np;
        int u = wv;
        for (int jb = 0; jb < np; jb++)
        {
            const int rbeg = 6 * jb, rend = isplit;
            const int nch = (rend - rbeg + 63) >> 6;
            for (; u < nch; u += nwv)
                ea_chunk(U, ldc, nru, nbr, rel, Ls, (long)LLD, 0L, jb, u, lane, sink, sink_lds, rbeg, rend);
            u -= nch;
        }
        __syncthreads(); // the next child may touch the same entries
    }
}

__device__ __forceinline__ int pad16(int nc) { return (nc + 15) & ~15; }

// load base[idx] with a 32-bit BYTE offset: `uniform 64-bit base + zero-extended 32-bit offset` is
// the addressing mode of global_load (saddr + voffset) — one VALU instruction per address instead
// of a 64-bit multiply-add pair.  idx * 8 must fit 32 bits (a front / a W block always does).
__device__ __forceinline__ double ldg32(const double* __restrict__ base, unsigned idx)
{
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + (idx << 3));
}

// lower triangle of F11 -> LDS (upper part zero), padded to a multiple of 16 with an identity
// block (W = L11^-1 is built from 16-column blocks).  The LDS copy always has leading dimension
// LLD = 113 (odd: column walks are conflict-free) whatever nc is, so that every LDS address in
// the factorisation is `base + compile-time offset` — one instruction per access.
__device__ __forceinline__ void l11_issue(const double* __restrict__ F, long ld, int nc, double (&v)[3][3])
{
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const unsigned uld = (unsigned)ld;
#pragma unroll
    for (int u = 0; u < 3; u++)
#pragma unroll
        for (int q = 0; q < 3; q++)
        {
            const int c = ty + 32 * u, r = tx + 32 * q;
            // (q < u: rows 32q.. lie above columns 32u.. for every thread — nothing of the lower triangle)
            v[u][q] = q < u ? 0.0 : ldg32(F, (unsigned)min(c, nc - 1) * uld + (unsigned)min(r, nc - 1));
        }
}
// MIRROR: the diagonal 16x16 tiles are stored symmetric right away (what dev_potrf16 otherwise does in a pass of
// its own after the extend-add: a front without children to add needs no such pass)
template <bool MIRROR = false>
__device__ __forceinline__ void l11_store(int nc, const double (&v)[3][3], double* __restrict__ Ls)
{
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int u = 0; u < 3; u++)
#pragma unroll
        for (int q = 0; q < 3; q++)
        {
            const int c = ty + 32 * u, r = tx + 32 * q;
            const double x = (c < nc && r < nc) ? (r >= c ? v[u][q] : 0.0) : (r == c ? 1.0 : 0.0);
            const bool tile = MIRROR && (r >> 4) == (c >> 4);
            if (!(tile && r < c)) // (the strict upper triangle of a diagonal tile is written by the mirror partner)
                Ls[c * LLD + r] = x;
            if (tile && r > c)
                Ls[r * LLD + c] = x;
        }
}
template <bool MIRROR = false>
__device__ __forceinline__ void dev_load_l11(const double* __restrict__ F, long ld, int nc, double* __restrict__ Ls)
{
    // 1024 threads cover the full NC_MAX x NC_MAX LDS matrix, 9 elements each: all global loads (6 per
    // thread: clamped addresses, no branch) are issued before the first LDS store; everything outside
    // the nc x nc lower triangle becomes identity / zero
    double v[3][3];
    l11_issue(F, ld, nc, v);
    l11_store<MIRROR>(nc, v, Ls);
}

// Factor the 6-column panel starting at (j0,j0) of the LDS matrix Ls by ONE wave, in registers.
// Every lane holds the 6x6 diagonal block (broadcast LDS reads) and factors it REDUNDANTLY —
// uniform values, no cross-lane traffic on the critical path — while applying each finished
// column to its own rows below the block (lane t: rows j0+6+t and j0+70+t; nc <= 96).
// Right-looking inside the panel, so every value is touched by one FMA per column: the
// dependent chain is 6 x (sqrt || 1/d, mul, mul, fma) ~ 6 x 220 cycles.  Writes dinv[j0..j0+5].
// The factored 6x6 diagonal block and the reciprocal pivots are NOT stored here: nothing reads them
// before the W phase, so the caller stores them (panel_store_diag) after the barrier that releases the
// other waves — 27 lane-0 LDS stores less on the one wave whose instruction stream is the chain.
template <bool HAS1>
__device__ __forceinline__ bool panel_factor_wave(double* __restrict__ Ls, int nc, int j0,
                                                  double* __restrict__ dinv, double (&Dout)[21], double (&ivout)[PW])
{
    const int lane = threadIdx.x & 63;
    const int r0 = j0 + PW + lane, r1 = j0 + PW + 64 + lane;
    // rows past the end are clamped to the last padded row for the loads (valid LDS, results
    // unused) and masked as a group for the stores: no branch per access
    const bool ok0 = r0 < nc, ok1 = HAS1 && r1 < nc;
    double* P = Ls + j0 * LLD + j0;           // (j0, j0); column c at P + c*LLD (immediate offsets)
    const double* A0 = Ls + j0 * LLD + min(r0, NC_MAX - 1);
    const double* A1 = Ls + j0 * LLD + min(r1, NC_MAX - 1);
    if (j0 == 0)
        stamp(2, 0);
    double D[PW][PW], a0[PW], a1[PW];
#pragma unroll
    for (int c = 0; c < PW; c++)
    {
#pragma unroll
        for (int r = 0; r < PW; r++)
            D[r][c] = (r >= c) ? P[c * LLD + r] : 0.0;
        a0[c] = A0[c * LLD];
        a1[c] = HAS1 ? A1[c * LLD] : 0.0;
    }
    bool bad = false;
    double iv[PW];
    if (j0 == 0)
    {
        // make the loads complete before the stamp: consume one value
        asm volatile("" ::"v"(D[PW - 1][PW - 1]), "v"(a1[PW - 1]));
        stamp(2, 1);
    }
#pragma unroll
    for (int j = 0; j < PW; j++)
    {
        const double d = D[j][j];
        // a pivot <= tol (or NaN) only raises the flag: whatever flows on (NaN from rsq of a
        // negative number included) is discarded with the rejected LM trial
        bad = bad || !(d > PIVOT_TOL);
        // sqrt(d) and 1/sqrt(d) together from one v_rsq_f64 seed (2^-24) and two coupled Newton
        // steps (the scheme of the IEEE sqrt expansion, without its range scaling and special-case
        // handling: d is a pivot in (1e-14, huge)): 11 instructions instead of ~28 for sqrt + 1/d
        double sq, inv;
        {
            const double y = __builtin_amdgcn_rsq(d);
            double g = d * y, h = 0.5 * y;
            const double r0 = fma(-h, g, 0.5);
            // h2 = 2h is carried along so that 1/sqrt(d) needs no final doubling on the chain
            const double h2 = fma(y, r0, y);
            g = fma(g, r0, g), h = fma(h, r0, h);
            const double dg = fma(-g, g, d), rh = fma(-h, g, 0.5);
            sq = fma(dg, h, g);
            inv = fma(h2, rh, h2);
        }
        iv[j] = inv;
        D[j][j] = sq;
#pragma unroll
        for (int i = 0; i < PW; i++)
            if (i > j)
                D[i][j] *= inv;
        a0[j] *= inv;
        if (HAS1)
            a1[j] *= inv;
#pragma unroll
        for (int c = 0; c < PW; c++)
            if (c > j)
            {
#pragma unroll
                for (int i = 0; i < PW; i++)
                    if (i >= c)
                        D[i][c] -= D[i][j] * D[c][j];
                a0[c] -= a0[j] * D[c][j];
                if (HAS1)
                    a1[c] -= a1[j] * D[c][j];
            }
    }
    if (j0 == 0)
    {
        asm volatile("" ::"v"(D[PW - 1][PW - 1]), "v"(a1[PW - 1]), "v"(a0[PW - 1]));
        stamp(2, 2);
    }
    // Branch-free stores: lanes without a row write into a sink (the Vs area behind dinv, unused
    // until the panels are done; offsets up to 127 + 5*LLD stay inside it).  With the stores under
    // `if (ok0)` the compiler sinks the loads and the whole update chain of a0/a1 into that
    // branch, i.e. behind the factorisation of D instead of into its latency shadows.
    {
        double* sink = dinv + NC_MAX + lane;
        double* W0 = ok0 ? Ls + j0 * LLD + r0 : sink;
#pragma unroll
        for (int c = 0; c < PW; c++)
            W0[c * LLD] = a0[c];
        if (HAS1)
        {
            double* W1 = ok1 ? Ls + j0 * LLD + r1 : sink + 64;
#pragma unroll
            for (int c = 0; c < PW; c++)
                W1[c * LLD] = a1[c];
        }
    }
    {
        int k = 0;
#pragma unroll
        for (int c = 0; c < PW; c++)
        {
            ivout[c] = iv[c];
#pragma unroll
            for (int r = 0; r < PW; r++)
                if (r >= c)
                    Dout[k++] = D[r][c];
        }
    }
    if (j0 == 0)
        stamp(2, 3);
    return bad;
}

__device__ __forceinline__ void panel_store_diag(double* __restrict__ Ls, int j0, double* __restrict__ dinv,
                                                 const double (&Dv)[21], const double (&iv)[PW])
{
    if ((threadIdx.x & 63) == 0)
    {
        double* P = Ls + j0 * LLD + j0;
        int k = 0;
#pragma unroll
        for (int c = 0; c < PW; c++)
        {
            dinv[j0 + c] = iv[c];
#pragma unroll
            for (int r = 0; r < PW; r++)
                if (r >= c)
                    P[c * LLD + r] = Dv[k++];
        }
    }
}

// Trailing update on the matrix cores: columns [c0, NC) of the LDS matrix -= P P^T with P the
// 6-column panel at j0, in 16x16 tiles dealt to `nw` waves (widx = index of this wave, < 0: not a
// member).  K = 6 is two v_mfma_f64_16x16x4 steps (k = 4,5 in the second, upper lanes zero).
// About 20 instructions per 256 elements instead of ~14 per element with vector FMAs: a wave
// issues roughly one instruction per 8 cycles, so instruction count is what matters here.
// Diagonal tiles also write their upper half (never read); rows/columns >= NC_MAX are masked.
__device__ __forceinline__ void panel_update_mfma(double* __restrict__ Ls, int nc, int j0, int c0,
                                                  int widx_, int nw)
{
    // the wave index as a scalar: tile coordinates and the edge test then live in SGPRs, and the common
    // case — a tile wholly inside the NC_MAX square — runs without a single per-element condition (an
    // exec-mask region per conditional store costs ~14 instructions; a wave's tile took 1.6 k cycles,
    // longer than the panel factorisation it is supposed to hide behind)
    const int widx = __builtin_amdgcn_readfirstlane(widx_);
    if (widx < 0 || c0 >= nc)
        return;
    const int lane = threadIdx.x & 63, ln = lane & 15, lk = lane >> 4;
    const int nt = (nc - c0 + 15) >> 4;
    const int ntiles = nt * (nt + 1) / 2;
    // K = PW columns of the panel: lane group lk supplies k = lk (and k = 4 + lk in a second MFMA
    // when PW > 4); lanes whose k is beyond the panel contribute zeros
    const double* P0 = Ls + (j0 + min(lk, PW - 1)) * LLD;
    const double* P1 = Ls + (j0 + min(4 + lk, PW - 1)) * LLD;
    const bool k0ok = lk < PW, k1ok = 4 + lk < PW;
    for (int t = widx; t < ntiles; t += nw)
    {
        int tj = 0, rem = t; // unrank: tile column tj has nt - tj tiles
        while (rem >= nt - tj)
        {
            rem -= nt - tj;
            tj++;
        }
        const int R = c0 + 16 * (tj + rem), C = c0 + 16 * tj;
        if (C + 16 <= NC_MAX) // scalar branch; rows may run into the padding below row NC_MAX (LLD)
        {
            const double a0 = k0ok ? -P0[R + ln] : 0.0, b0 = P0[C + ln];
            const double a1 = k1ok ? -P1[R + ln] : 0.0, b1 = P1[C + ln];
            double4_t acc;
            double* Cc = Ls + (C + ln) * LLD + R + lk;
#pragma unroll
            for (int q = 0; q < 4; q++)
                acc[q] = Cc[4 * q];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
            if (PW > 4)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; q++)
                Cc[4 * q] = acc[q];
            continue;
        }
        const int ra = min(R + ln, NC_MAX - 1), cb = min(C + ln, NC_MAX - 1);
        const double a0 = k0ok ? -P0[ra] : 0.0, b0 = P0[cb];
        const double a1 = k1ok ? -P1[ra] : 0.0, b1 = P1[cb];
        double4_t acc;
        double* Cc = Ls + cb * LLD;
#pragma unroll
        for (int q = 0; q < 4; q++)
            acc[q] = Cc[min(R + lk + 4 * q, NC_MAX - 1)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
        if (PW > 4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
        if (C + ln < NC_MAX)
        {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (R + lk + 4 * q < NC_MAX)
                    Cc[R + lk + 4 * q] = acc[q];
        }
    }
}

// the PW columns right after panel j0 (the next panel), one matrix element per thread:
// column = t >> 7, row = j0 + PW + (t & 127), t = tid - 256   (needs blockDim >= 256 + 128 * PW, nc <= 128)
__device__ __forceinline__ void panel_update_next(double* __restrict__ Ls, int nc, int j0)
{
    // threads 256 .. 1023: the first wave stores the previous diagonal block meanwhile (dev_potrf_panels)
    const int t = (int)threadIdx.x - 256;
    const int cc = t >> 7, rr = t & 127;
    const int c = j0 + PW + cc, r = j0 + PW + rr;
    if (t >= 0 && cc < PW && c < nc && r < nc && r >= c)
    { // two partial sums: a dependent fp64 FMA costs ~40 cycles
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < PW; k++)
        {
            const double p = Ls[(j0 + k) * LLD + r] * Ls[(j0 + k) * LLD + c];
            if (k & 1)
                s1 += p;
            else
                s0 += p;
        }
        Ls[c * LLD + r] -= s0 + s1;
    }
}

// L11 = chol(F11) in LDS (Ls: nc x nc, leading dimension LLD), PW-column panels with
// LOOK-AHEAD: once panel p is factored, all threads first update only the PW columns of panel
// p+1; then wave 0 factors panel p+1 in registers while the other waves apply panel p to the
// remaining columns on the matrix cores.  Two barriers per panel, and the sequential part (panel
// factorisation) overlaps the parallel part (trailing update).
// On return Ls holds L11 (lower) and dinv the reciprocal diagonal.  L11 is NOT written back to
// F: every later consumer (trsm, backward substitution) works with W = L11^-1 (dev_winv).
template <bool MIRROR = false>
__device__ __forceinline__ void dev_potrf_load(const double* __restrict__ F, long ld, int nc, double* __restrict__ Ls,
                                               double* __restrict__ dinv)
{
    dev_load_l11<MIRROR>(F, ld, nc, Ls);
    if (threadIdx.x < NC_MAX)
        dinv[threadIdx.x] = 1.0; // identity padding; the panels overwrite the real columns
    __syncthreads();
}

__device__ __forceinline__ void dev_potrf_panels(int nc, double* __restrict__ Ls, double* __restrict__ dinv,
                                                 int32_t* __restrict__ fail);

__device__ __forceinline__ void dev_potrf_panels(int nc, double* __restrict__ Ls, double* __restrict__ dinv,
                                                 int32_t* __restrict__ fail)
{
    stamp(0, 2);
    double Dv[21], iv[PW]; // first wave: the diagonal block of the panel factored last, stored one phase later
    if (threadIdx.x < 64)
        if (nc > PW + 64 ? panel_factor_wave<true>(Ls, nc, 0, dinv, Dv, iv) : panel_factor_wave<false>(Ls, nc, 0, dinv, Dv, iv))
            *fail = 1;
    __syncthreads();
    stamp(0, 3);
    int jlast = 0;
    for (int j0 = 0; j0 < nc; j0 += PW)
    {
        const int jn = j0 + PW;
        if (jn >= nc)
            break;
        if (threadIdx.x < 64)
            panel_store_diag(Ls, j0, dinv, Dv, iv); // off the chain: the others update the next panel's columns
        panel_update_next(Ls, nc, j0); // one element per thread (threads 256 ..)
        __syncthreads();
        if (threadIdx.x < 64)
        {
            // rows beyond the first 64 below the panel (second register row per lane) exist only
            // in the first panels of a wide pivot block
            if (jn + PW + 64 < nc ? panel_factor_wave<true>(Ls, nc, jn, dinv, Dv, iv)
                                  : panel_factor_wave<false>(Ls, nc, jn, dinv, Dv, iv))
                *fail = 1;
        }
        else
            panel_update_mfma(Ls, nc, j0, jn + PW, (int)(threadIdx.x >> 6) - 1, (int)(blockDim.x >> 6) - 1); // the rest, meanwhile
        __syncthreads();
        jlast = jn;
    }
    if (threadIdx.x < 64)
        panel_store_diag(Ls, jlast, dinv, Dv, iv);
    stamp(0, 4);
}

// ---------------------------------------------------------------- 16-column panels (default) ----
// k_up_potrf, round 3.  The 6-column panel loop above is bound by ONE wave's instruction stream (a
// wave issues at most one instruction per ~4-5 cycles, whatever the instruction): per 6 columns it
// pays ~100 LDS accesses, two barriers and a redundant 6x6 factorisation in every lane, and W = L11^-1
// is then built in a phase of its own (12 k cycles) after the last panel.  This form:
//   * keeps a 16-column panel in REGISTERS, lane <-> row, with no LDS access on the chain: the matrix
//     is symmetric, so lane k of the 16 lanes that hold the rows of the diagonal block also holds (in
//     its registers j > k) ROW k of the trailing block — exactly the multipliers step k needs.  They
//     are broadcast with v_readlane (to SGPRs; an FMA takes one SGPR operand), not through LDS;
//   * factors L D L^T, not L L^T: step k is  d = a[k][k] (lane k);  y = 1/d (v_rcp + two Newton steps:
//     5 instructions where sqrt || rsqrt needs 10);  a[i][j] -= (a[i][k] y) a[k][j]  for j > k — one FMA
//     per column, rows below the diagonal block ride along for free (same instructions, other lanes).
//     The panel is stored UNSCALED (a[i][k] = l_ik d_k) with the reciprocal pivots beside it; every
//     consumer scales the operand it loads (one multiply per loaded value, off the critical wave);
//   * 16 columns = 136 values to broadcast (272 v_readlane_b32) + 120 FMAs + 16 x 8 for the pivots:
//     ~490 instructions per 16 columns where the 6-column form needs ~3 x 330 plus its LDS round trips;
//   * K = 16 trailing updates are four v_mfma_f64_16x16x4 per 16x16 tile (two independent chains): the
//     tiles of the NEXT panel's block column first (phase A: <= 5 tiles, one wave each), then the panel
//     waves factor it while the remaining tiles are updated (phase B).  Diagonal tiles are kept
//     symmetric in LDS (mirrored once after the load, updated as full tiles), so a panel load is 16
//     plain ds_reads with immediate offsets;
//   * W = L11^-1 = D^-1/2 U, U = L^-1 (unit lower), is built WHILE the panels run, by the waves the
//     trailing update leaves idle: in the slot after panel J is factored one wave inverts its diagonal
//     block (V_J, 4 lanes per column), checks its pivots (zero-pivot flag) and forms d^-1/2; one slot
//     later block row J of U follows on the matrix cores, U_IJ = -V_I sum_K L_IK U_KJ (earlier U blocks
//     parked in LDS).  After the last panel only V of the last block and one block row remain.
// Rows 64.. of a panel (the first two panels of a 96-column block) go to a second wave that carries
// the 16 rows of the diagonal block redundantly in its lanes 0..15: same instruction stream, another SIMD.
// quad-broadcast of lane Q of every quad (DPP quad_perm: two v_mov_dpp, no LDS)
template <int Q>
__device__ __forceinline__ double quad_bcast(double x)
{
    constexpr int ctrl = Q * 0x55; // quad_perm [Q, Q, Q, Q]
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

namespace p16
{
constexpr int OFF_INV = NC_MAX * LLD;     // reciprocal pivots 1/d            [96]
constexpr int OFF_RS = OFF_INV + NC_MAX;  // d^-1/2                            [96]
constexpr int OFF_UB = OFF_RS + NC_MAX;   // blocks of U = L^-1 (unit lower), block (I,J), J <= I, at I(I+1)/2 + J, [k*16 + n]
constexpr int LDS_DOUBLES = OFF_UB + 21 * 256;
} // namespace p16

__device__ __forceinline__ double readlane_f64(double x, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// the lane index behind an optimisation barrier: the address arithmetic of a role (panel / tile / U block)
// is then redone where it is used instead of being hoisted out of the slot loop, where the hoisted
// values of ALL roles stay live side by side and push the kernel over its 128 VGPRs (scratch spills)
__device__ __forceinline__ int opaque_lane()
{
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    return lane;
}

// diagnosis (CUGO_DEBUG_DELAY=n): chosen waves / workgroups of k_up_potrf sleep ~25 k cycles at chosen points.  A
// kernel without a race gives the same bits whatever runs late.
// Only in the hooks build (make HOOKS=1): in the product build DBG_DELAY is the constant 0 and every pattern folds away.
#ifdef CUGO_DEBUG_HOOKS
#define DBG_DELAY(p) ((p).dbg_delay)
#else
#define DBG_DELAY(p) 0
#endif
__device__ __forceinline__ void dbg_sleep()
{
#pragma unroll 1
    for (int i = 0; i < 3; i++)
        __builtin_amdgcn_s_sleep(127);
}
// The other kernels of the factorisation — trsm / syrk / fused tiles / backward substitution — hand over between their
// phases through barriers only.  CUGO_DEBUG_DELAY=10..14 (hooks build) puts chosen waves to sleep BEHIND EVERY ONE of
// those barriers (10: odd waves, 11: even waves, 12: waves 2 3 6 7 ..., 13: wave 0 alone, 14: all but wave 0): a
// hand-over that relies on two waves leaving a barrier at about the same time gives other bits under one of them.
#ifdef CUGO_DEBUG_HOOKS
__device__ int g_dbg_tile_delay = 0; // (set by the launch functions from CholPlanDev::dbg_delay)
__device__ __forceinline__ void dbg_tile_delay()
{
    const int d = g_dbg_tile_delay;
    if (d < 10)
        return;
    const int w = (int)(threadIdx.x >> 6);
    if ((d == 10 && (w & 1)) || (d == 11 && !(w & 1)) || (d == 12 && (w & 2)) || (d == 13 && w == 0) || (d == 14 && w != 0))
        dbg_sleep();
}
#define TILE_SYNC()        \
    do                     \
    {                      \
        __syncthreads();   \
        dbg_tile_delay();  \
    } while (0)
#else
#define TILE_SYNC() __syncthreads()
#endif

__device__ __forceinline__ void stamp_wave(int kernel, int slot)
{
#ifdef CUGO_STAMPS
    if (g_stamps && (int)blockIdx.x == g_stamp_block[kernel] && (threadIdx.x & 63) == 0)
        g_stamps[kernel * 8 + slot] = clock64();
#else
    (void)kernel, (void)slot;
#endif
}

// half 0: rows j0 .. j0+63; half 1: the diagonal block's rows again (lanes 0..15, not stored) and rows
// j0+64 .. j0+95 (lanes 16..47).  ncp = padded size of the LDS matrix (identity beyond nc).
// When a panel has a second half (ncp - j0 > 64) the two waves run side by side with no synchronisation between
// them, and half 1 LOADS the diagonal tile that half 0 factors in place.  Half 0 therefore parks the 16 rows of the
// diagonal tile in `stage` (column stride LLD) instead of storing them over the tile, and the tile is filled in from
// there behind the next barrier (p16_unstage).  (Stored in place, a second wave that issued its loads late — behind
// the ~200 operand loads the other 14 waves queue after the barrier — read a tile whose first columns were already
// eliminated: a slightly wrong L for ITS rows, i.e. block row 5 of a 96-column front, about once in 10^5 such
// fronts: the run-to-run deviation of DESIGN.md section 2.)
__device__ __forceinline__ void panel16_factor(double* __restrict__ Ls, int ncp, int j0, double* __restrict__ invd,
                                               int half, double* __restrict__ stage, bool park = true)
{
    const int lane = opaque_lane();
    const int rr = half == 0 ? lane : (lane < 16 ? lane : lane + 48); // row inside the panel
    // rows past the matrix are clamped into the 16 padding rows below it for the loads (valid LDS,
    // values unused: nothing is ever broadcast from these lanes) and stored into those rows
    const int rrc = min(rr, NC_MAX + 15 - j0);
    const double* Pc = Ls + j0 * LLD + j0 + rrc;
    if (j0 == 32 && half == 0)
        stamp(2, 0);
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; c++)
        a[c] = Pc[c * LLD]; // (rows of the diagonal block: its tile is symmetric in LDS)
    if (j0 == 32 && half == 0)
    {
        asm volatile("" ::"v"(a[15]), "v"(a[0]), "v"(a[7]));
        stamp(2, 1);
    }
    const bool st = (half == 0 || lane >= 16) && j0 + rr < ncp;
    const bool parked = park && half == 0 && lane < 16 && ncp - j0 > 64; // (the diagonal tile's rows while half 1 may read it)
    double* W0 = parked ? stage + lane : Ls + j0 * LLD + (st ? j0 + rr : NC_MAX + (lane & 15));
    double iv[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
        const double d = readlane_f64(a[k], k);
        // 1/d: v_rcp_f64 seed and two Newton steps.  A pivot <= tol (or NaN) is found by the wave that
        // inverts this diagonal block (it raises the flag): whatever flows on is discarded with the trial
        double y = __builtin_amdgcn_rcp(d);
        // row k of the trailing block (lane k's registers j > k) to SGPRs: independent of the reciprocal,
        // issued in the shadow of v_rcp_f64.  The scheduling barriers keep this order — left alone, the
        // scheduler of the loop instance re-uses ONE SGPR pair and pays a hazard s_nop per FMA
        double sk[16];
#pragma unroll
        for (int j = k + 1; j < 16; j++)
            sk[j] = readlane_f64(a[j], k);
        // column k is final (unscaled: l_ik d_k; the diagonal entry is d_k): it leaves for LDS now, so that
        // at the end of the panel only the last store is still in flight (16 stores issued together at the
        // end wait ~1000 cycles in the LDS queue behind the other waves' traffic)
        W0[k * LLD] = a[k];
        __builtin_amdgcn_sched_barrier(0);
        double e = fma(-d, y, 1.0);
        y = fma(y, e, y);
        e = fma(-d, y, 1.0);
        y = fma(y, e, y);
        iv[k] = y;
        const double t = a[k] * y; // l_ik
#pragma unroll
        for (int j = k + 1; j < 16; j++)
            a[j] = fma(-t, sk[j], a[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (j0 == 32 && half == 0)
    {
        asm volatile("" ::"v"(a[15]), "v"(a[0]), "v"(a[7]));
        stamp(2, 2);
    }
    if (half == 0 && lane == 0)
    {
        double* dv = invd + j0;
#pragma unroll
        for (int k = 0; k < 16; k++)
            dv[k] = iv[k];
    }
    if (j0 == 32 && half == 0)
        stamp(2, 3);
}

// the parked rows of the diagonal tile at j0 (panel16_factor) into their place: one wave, behind a barrier
__device__ __forceinline__ void p16_unstage(double* __restrict__ Ls, const double* __restrict__ stage, int j0)
{
    const int lane = opaque_lane(), r = lane & 15, k4 = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; q++)
        Ls[(j0 + k4 + 4 * q) * LLD + j0 + r] = stage[(k4 + 4 * q) * LLD + r];
}

// trailing tile (rows R.., columns C..) -= L_R D L_C^T, from the factored 16-column panel at j0 (stored
// unscaled: l_ik d_k, reciprocal pivots in invd)
__device__ __forceinline__ void panel16_update_tile(double* __restrict__ Ls, const double* __restrict__ invd, int j0,
                                                    int R, int C)
{
    const int lane = opaque_lane(), ln = lane & 15, lk = lane >> 4;
    const double* Pk = Ls + (j0 + lk) * LLD;
    double a[4], b[4], y[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
        y[kk] = invd[j0 + lk + 4 * kk], a[kk] = Pk[4 * kk * LLD + R + ln], b[kk] = Pk[4 * kk * LLD + C + ln];
    double* Cc = Ls + (C + ln) * LLD + R + lk;
    double c[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
        c[q] = Cc[4 * q];
    double4_t s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0] * y[0], b[0], s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1] * y[1], b[1], s1, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2] * y[2], b[2], s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3] * y[3], b[3], s1, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++)
        Cc[4 * q] = c[q] - (s0[q] + s1[q]);
}

// x := L^-1 x for the unit lower 16 x 16 block at Lb (Lb already offset by r4; l_ik = Lb[k*LLD + 4m] / d_k,
// yv = 1/d), quad layout (see p16_utask).  Step K: rows below K take  x_i -= l_iK x_K.
template <int K>
__device__ __forceinline__ void p16_sub_step(const double* __restrict__ Lb, int r4, const double (&c)[4], double yk,
                                             double (&x)[4], double (&cn)[4], double& yn, const double* __restrict__ yv)
{
    constexpr int ko = K & 3, km = K >> 2;
    // the coefficients of the NEXT step first: their latency hides behind this step's chain
    if (K + 1 < 15)
    {
        constexpr int kmn = (K + 1) >> 2;
#pragma unroll
        for (int m = kmn; m < 4; m++)
            cn[m] = Lb[(K + 1) * LLD + 4 * m];
        yn = yv[K + 1];
    }
    __builtin_amdgcn_sched_barrier(0);
    const double xk = quad_bcast<ko>(x[km]) * yk;
#pragma unroll
    for (int m = km; m < 4; m++)
    { // rows below K only: for m == km that is r4 > ko
        const double coef = (m > km || r4 > ko) ? c[m] : 0.0;
        x[m] = fma(-coef, xk, x[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void p16_substitute(const double* __restrict__ Lb, const double* __restrict__ yv, int r4,
                                               double (&x)[4])
{
    double c[4], cn[4] = {0, 0, 0, 0}, yk = yv[0], yn = 0.0;
#pragma unroll
    for (int m = 0; m < 4; m++)
        c[m] = Lb[4 * m];
#define P16_STEP(K)                                   \
    p16_sub_step<K>(Lb, r4, c, yk, x, cn, yn, yv);    \
    c[0] = cn[0], c[1] = cn[1], c[2] = cn[2], c[3] = cn[3], yk = yn;
    P16_STEP(0) P16_STEP(1) P16_STEP(2) P16_STEP(3) P16_STEP(4) P16_STEP(5) P16_STEP(6) P16_STEP(7)
    P16_STEP(8) P16_STEP(9) P16_STEP(10) P16_STEP(11) P16_STEP(12) P16_STEP(13) P16_STEP(14)
#undef P16_STEP
}

// One wave: block (I, J), J <= I, of U = L^-1 (unit lower):  L_II U_IJ = -sum_{K=J..I-1} L_IK U_KJ  (J < I),
// L_II U_II = I.  The sum runs on the matrix cores (four independent accumulators: a chain of I - J MFMAs),
// goes through the block's LDS slot into the substitution layout, and the 16 x 16 unit lower triangular
// solve is a forward substitution in registers (p16_substitute): no explicit inverse of the diagonal block,
// no dependent MFMA chain behind it.  The block is parked in LDS for the rows below and stored to global
// memory as W_IJ = D_I^-1/2 U_IJ.  The task of the diagonal block also checks the pivots of block row I
// (zero-pivot flag; ref: csrcholZeroPivot tol).
// (Measured alternatives, all within 0.15 ms of each other per 10-iteration step: an explicit inverse V_I by
// one wave and U_IJ = -V_I S one slot later; the sums kept right-looking, each term added by the task that
// formed U_KJ, or by the idle waves of the next phase A — every form moves the same work between waves of
// ONE CU that share its LDS pipe and issue slots.)
__device__ __forceinline__ void p16_utask(const double* __restrict__ Ls, const double* __restrict__ invd,
                                          double* __restrict__ rsv, double* __restrict__ Ub, int I, int J,
                                          double* __restrict__ Wg, int ncp, int32_t* __restrict__ fail)
{
    const int lane = opaque_lane(), ln = lane & 15, lk = lane >> 4;
    double* slot = Ub + (I * (I + 1) / 2 + J) * 256;
    if (J < I)
    {
        double4_t s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, s3 = {0, 0, 0, 0};
        for (int K = J; K < I; K++)
        {
            // A = L_IK [m = ln][k = lk + 4kk] = Ls[col 16K + k][row 16I + m] / d_k ;  B = U_KJ [k][n = ln]
            const double* La = Ls + (16 * K + lk) * LLD + 16 * I + ln;
            const double* Bb = Ub + (K * (K + 1) / 2 + J) * 256 + lk * 16 + ln;
            double a[4], b[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
                a[kk] = La[4 * kk * LLD] * invd[16 * K + lk + 4 * kk], b[kk] = Bb[64 * kk];
            s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], s2, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], s3, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) // result element [m = lk + 4q][n = ln]
            slot[(lk + 4 * q) * 16 + ln] = -((s0[q] + s1[q]) + (s2[q] + s3[q]));
    }
    else
    { // the diagonal block: right-hand side = identity (through the slot as well: one code path below)
#pragma unroll
        for (int q = 0; q < 4; q++)
            slot[(lk + 4 * q) * 16 + ln] = (lk + 4 * q == ln) ? 1.0 : 0.0;
    }
    wave_lds_sync();
    // forward substitution, four lanes per column: lane = 4 n + r4 holds rows r4 + 4m of column n.  The one
    // value a step shares — the finished entry x_k — goes round the quad by DPP (no LDS); a lane needs
    // <= 4 coefficients per step, loaded one step ahead (broadcast reads: they depend on r4 only)
    const int n = lane >> 2, r4 = lane & 3;
    double x[4];
#pragma unroll
    for (int m = 0; m < 4; m++)
        x[m] = slot[(r4 + 4 * m) * 16 + n];
    const double* Lb = Ls + (16 * I) * LLD + 16 * I + r4;
    const double* yv = invd + 16 * I;
    {
        // d^-1/2 of block row I (lane k forms entry k; every task of the row writes the same values)
        const double d = Ls[(16 * I + ln) * LLD + 16 * I + ln];
        const double y = __builtin_amdgcn_rsq(d);
        double g = d * y, h = 0.5 * y;
        const double r0 = fma(-h, g, 0.5);
        const double h2 = fma(y, r0, y);
        g = fma(g, r0, g), h = fma(h, r0, h);
        const double rh = fma(-h, g, 0.5);
        if (lane < 16)
            rsv[16 * I + ln] = fma(h2, rh, h2);
        if (J == I && __ballot(!(d > PIVOT_TOL)) != 0 && lane == 0) // (a NaN pivot counts)
            *fail = 1;
    }
    p16_substitute(Lb, yv, r4, x);
    wave_lds_sync(); // rsv (and the slot: every lane has read its entries)
    double* wg = Wg + (long)(16 * J + n) * ncp + 16 * I + r4;
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
        slot[(r4 + 4 * m) * 16 + n] = x[m];
        wg[4 * m] = x[m] * rsv[16 * I + r4 + 4 * m];
    }
}

// F11 (ncp x ncp in LDS, lower triangle, identity padding) -> W = F11^-1/2-inverse in global memory.
// lds: Ls | invd | rsv | Vs | Ub  (p16::LDS_DOUBLES doubles)
__device__ __forceinline__ void dev_potrf16(int ncp, double* __restrict__ lds, double* __restrict__ Wg,
                                            int32_t* __restrict__ fail, bool mirrored = false, int dbg_delay = 0)
{
    double* Ls = lds;
    double* invd = lds + p16::OFF_INV;
    double* rsv = lds + p16::OFF_RS;
    double* Ub = lds + p16::OFF_UB;
    const int nblk = ncp >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // the diagonal 16x16 tiles become symmetric: the trailing update treats them as full tiles and a
    // panel load is then the same plain column walk for every lane
    if (!mirrored) // (uniform; a front without children to add had its tiles mirrored by the load)
    {
        for (int e = threadIdx.x; e < nblk * 256; e += blockDim.x)
        {
            const int b = e >> 8, r = (e >> 4) & 15, c = e & 15;
            if (r > c)
                Ls[(16 * b + r) * LLD + 16 * b + c] = Ls[(16 * b + c) * LLD + 16 * b + r];
        }
        __syncthreads();
    }
    stamp(0, 2);
    // parking place of a two-wave panel's diagonal tile: the slots of U from block (1,0) on — nothing is written
    // there before phase B of slot 1, and the last two-wave panel (j0 = 16) is put back in phase A of slot 1
    double* stage = Ub + 256;
    // (diagnosis: CUGO_DEBUG_DELAY=7 — the second panel wave alone runs late, the case the parking place is for;
    // =8 — the same WITHOUT the parking place, i.e. the kernel as it was: deviates at once)
    const bool park = dbg_delay != 8 && dbg_delay != 9; // (9: no parking place and no delay — the old kernel, to time the difference)
    if (w == 0 || (w == 1 && ncp > 64))
    {
        if ((dbg_delay == 7 || dbg_delay == 8) && w == 1)
            dbg_sleep();
        panel16_factor(Ls, ncp, 0, invd, w, stage, park);
    }
    __syncthreads();
    stamp(0, 3);
    // slot s: [phase A: block column s+1 updated by panel s] [phase B: panel s+1 factored || the other
    // block columns updated by panel s || block row s of U]; the last slot has no panel
    for (int s = 0; s < nblk; s++)
    {
        const int j0 = 16 * s, jn = j0 + 16;
        const bool panel = s + 1 < nblk;
        if (s == 1)
            stamp(5, 0);
        if (panel)
        {
            // phase A: the block column of the next panel, <= 5 tiles, one per wave, on waves 0 1 2 3 6 (four
            // different SIMDs first)
            const int ta = w < 4 ? w : (w == 6 ? 4 : 99);
            if (w == 7 && park && ncp - j0 > 64) // panel s had two waves: its diagonal tile comes out of the parking place
                p16_unstage(Ls, stage, j0);
            if (dbg_delay == 3 && (w & 1))
                dbg_sleep();
            // (hooks build, CUGO_DEBUG_DELAY=20..27: parts of the kernel LEFT OUT — wrong results, right timing; what
            // each part costs: profiles/r04_potrf_cost_by_omission.txt.  20: the tiles of phase A, 21: and its barrier,
            // 22: the U tasks, 23: the trailing tiles of phase B, 24: the panels of the slots, 25 = 22 + 23,
            // 26 = 22 + 24, 27 = 20 + 22 + 23 + 24)
            const bool skip_a = dbg_delay == 20 || dbg_delay == 21 || dbg_delay == 27 || dbg_delay == 28;
            if (!skip_a && s + 1 + ta < nblk)
                panel16_update_tile(Ls, invd, j0, 16 * (s + 1 + ta), jn);
            if (dbg_delay != 21)
                __syncthreads();
        }
        if (s == 1)
            stamp(5, 1);
        // wave 1 carries rows 64.. of the next panel (only the first panels of a 96-column block have them)
        const bool panel1 = panel && ncp - jn > 64;
        if ((dbg_delay == 1 && w >= 2) || (dbg_delay == 2 && w < 2) || (dbg_delay == 4 && (w & 2)))
            dbg_sleep();
        if ((w == 0 && panel) || (w == 1 && panel1))
        {
            // the panel wave is the critical path of the slot: it goes first wherever it competes with the
            // other waves of the CU (after the barrier every wave issues its LDS operand loads at once; at
            // equal priority the panel's 16 loads — and later its stores — queue behind ~200 others)
            __builtin_amdgcn_s_setprio(3);
            if ((dbg_delay == 7 || dbg_delay == 8) && w == 1)
                dbg_sleep();
            if (dbg_delay != 24 && dbg_delay != 26 && dbg_delay != 27 && dbg_delay != 28)
                panel16_factor(Ls, ncp, jn, invd, w, stage, park);
            __builtin_amdgcn_s_setprio(0);
            if (s == 1)
                stamp(5, 2);
        }
        else
        {
            // (measured, not kept: the other waves starting ~200 cycles later so that the panel's 16 loads sit at the
            // head of the CU's LDS queue — 11.55 vs 11.54 ms per step)
            // one task per wave: the s + 1 blocks of row s of U, then the remaining trailing tiles.
            // Waves w and w + 4 share a SIMD (observed placement; speed only): the tasks go first to the SIMDs
            // without a panel wave, the heavy ones (U blocks) first, and round the SIMDs:
            //   two panel waves (SIMDs 0, 1):  SIMDs 2, 3 (8 waves), then SIMD 1's, then SIMD 0's other waves
            //   one panel wave (SIMD 0):       SIMDs 1, 2, 3 (12 waves), then SIMD 0's other waves
            //   no panel (last slot):          every wave, SIMDs 0 1 2 3 0 1 ...
            int tw, ntw;
            if (!panel)
                tw = w, ntw = 16;
            else if (panel1)
                tw = (w & 3) >= 2 ? (w >> 2) * 2 + (w & 1) : ((w & 3) == 1 ? 8 + (w >> 2) - 1 : 11 + (w >> 2) - 1), ntw = 14;
            else
                tw = (w & 3) ? ((w & 3) - 1) + 3 * (w >> 2) : 11 + (w >> 2), ntw = 15;
            int t = tw;
            if (t <= s)
            {
                if (dbg_delay != 22 && dbg_delay != 25 && dbg_delay != 26 && dbg_delay != 27 && dbg_delay != 28)
                    p16_utask(Ls, invd, rsv, Ub, s, t, Wg, ncp, fail);
                if (s == 1 && t == 0)
                    stamp_wave(5, 4);
                if (s == 4 && t == 0)
                    stamp_wave(5, 5);
            }
            else if (s + 2 < nblk)
            {
                const int nt = nblk - (s + 2); // block columns still to update
                const int ntiles = nt * (nt + 1) / 2;
                for (t -= s + 1; t < ntiles; t += ntw - (s + 1))
                {
                    int tj = 0, rem = t;
                    while (rem >= nt - tj)
                    {
                        rem -= nt - tj;
                        tj++;
                    }
                    if (dbg_delay != 23 && dbg_delay != 25 && dbg_delay != 27 && dbg_delay != 28)
                        panel16_update_tile(Ls, invd, j0, 16 * (s + 2 + tj + rem), 16 * (s + 2 + tj));
                }
            }
        }
        __syncthreads();
        if (s == 1)
            stamp(5, 3);
        if (s == nblk - 2)
            stamp(0, 4);
    }
    stamp(0, 5);
}

// V = inverse of the 16x16 diagonal block `blk` of L11 (lower triangular) by 16 lanes: lane c
// builds column c by forward substitution with the block of L broadcast from LDS (fully
// unrolled: the v's stay in registers).  Vs[blk][n*17 + k] = V[n][k]; when Wg != nullptr the
// block is also written to the global W (column-major, leading dimension ncp).
__device__ __forceinline__ void inv_diag16_block(const double* __restrict__ Ls, const double* __restrict__ dinv,
                                                 int blk, int c, double* __restrict__ Vs,
                                                 double* __restrict__ Wg, int ncp)
{
    const double* Lb = Ls + (16 * blk) * LLD + 16 * blk;
    const double* di = dinv + 16 * blk; // reciprocal diagonal from the panel factorisation
    // right-looking forward substitution, in place: v[k] = e_c, then for k = 0..15
    // v[k] *= 1/L[k][k] and every later row is updated at once (independent FMAs), so the
    // dependent chain is 16 x (mul + fma), not 120 FMAs.  16 doubles of registers per lane.
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
        v[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
        v[k] *= di[k];
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (i > k)
                v[i] -= Lb[k * LLD + i] * v[k];
    }
    double* V = Vs + blk * (16 * 17);
#pragma unroll
    for (int i = 0; i < 16; i++)
        V[i * 17 + c] = (i >= c) ? v[i] : 0.0;
    if (Wg)
    {
        double* wc = Wg + (long)(16 * blk + c) * ncp + 16 * blk;
#pragma unroll
        for (int i = 0; i < 16; i++)
            wc[i] = (i >= c) ? v[i] : 0.0;
    }
}

// The same inverse by a WHOLE wave: four lanes per column (lane = 4 * column + r4), lane r4 of a quad
// holds the rows i = r4 + 4m of its column.  A substitution step is bound by the instructions a
// single wave can issue, not by its chain (16 lanes: 15 FMAs + 15 broadcast LDS loads per step for one
// wave, ~340 cycles): split four ways a step is <= 4 FMAs + 4 loads per lane plus ONE cross-lane move,
// the quad-broadcast of the scaled pivot-row entry (DPP quad_perm: two v_mov_dpp, no LDS).

template <int K>
__device__ __forceinline__ void inv_diag16_step(const double* __restrict__ Lb, const double* __restrict__ di, int r4,
                                                double (&v)[4])
{
    constexpr int ko = K & 3, km = K >> 2;
    const double vk = quad_bcast<ko>(v[km] * di[K]); // row K of this column, scaled: final
    v[km] = (r4 == ko) ? vk : v[km];
#pragma unroll
    for (int m = km; m < 4; m++)
    {
        // rows below K only: for m == km that is r4 > ko (clamped address, zero coefficient otherwise)
        const double l = Lb[K * LLD + r4 + 4 * m];
        const double coef = (m > km || r4 > ko) ? l : 0.0;
        v[m] -= coef * vk;
    }
}

__device__ __forceinline__ void inv_diag16_wave(const double* __restrict__ Ls, const double* __restrict__ dinv,
                                                int blk, double* __restrict__ Vs)
{
    const int lane = threadIdx.x & 63, r4 = lane & 3, c = lane >> 2;
    const double* Lb = Ls + (16 * blk) * LLD + 16 * blk;
    const double* di = dinv + 16 * blk;
    double v[4];
#pragma unroll
    for (int m = 0; m < 4; m++)
        v[m] = (r4 + 4 * m == c) ? 1.0 : 0.0;
    inv_diag16_step<0>(Lb, di, r4, v), inv_diag16_step<1>(Lb, di, r4, v), inv_diag16_step<2>(Lb, di, r4, v);
    inv_diag16_step<3>(Lb, di, r4, v), inv_diag16_step<4>(Lb, di, r4, v), inv_diag16_step<5>(Lb, di, r4, v);
    inv_diag16_step<6>(Lb, di, r4, v), inv_diag16_step<7>(Lb, di, r4, v), inv_diag16_step<8>(Lb, di, r4, v);
    inv_diag16_step<9>(Lb, di, r4, v), inv_diag16_step<10>(Lb, di, r4, v), inv_diag16_step<11>(Lb, di, r4, v);
    inv_diag16_step<12>(Lb, di, r4, v), inv_diag16_step<13>(Lb, di, r4, v), inv_diag16_step<14>(Lb, di, r4, v);
    inv_diag16_step<15>(Lb, di, r4, v);
    double* V = Vs + blk * (16 * 17);
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
        const int i = r4 + 4 * m;
        V[i * 17 + c] = (i >= c) ? v[m] : 0.0;
    }
}

// all diagonal blocks at once, one 16-lane group per block
__device__ __forceinline__ void dev_inv_diag16(const double* __restrict__ Ls, const double* __restrict__ dinv,
                                               int ncp, double* __restrict__ Vs)
{
    const int grp = threadIdx.x >> 4, c = threadIdx.x & 15;
    if (grp < (ncp >> 4))
        inv_diag16_block(Ls, dinv, grp, c, Vs, nullptr, ncp);
}

// W = L11^-1 (lower triangular, ncp x ncp with the identity padding) on the f64 matrix cores,
// written to global memory column-major: Wg[col*ncp + row].  Wave J builds block column J top
// down:  W_JJ = V_J,  W_IJ = -V_I * sum_{K=J..I-1} L_IK W_KJ.  The MFMA result layout
// (reg q: row (l>>4)+4q, col l&15) IS the B-operand layout of the next product, so the whole
// chain stays in registers: no LDS round trip between the dependent products.
// Everything after the factorisation works with W: trsm becomes a plain GEMM (no serial
// block substitution per tile) and the backward substitution a matrix-vector product (no
// 96-step serial solve).
__device__ __forceinline__ void dev_winv(const double* __restrict__ Ls, int ncp, const double* __restrict__ Vs,
                                         double* __restrict__ Wg)
{
    const int nblk = ncp >> 4;
    // (the wave index as a scalar: the block counts below are then scalar conditions, not exec-mask regions)
    const int J = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, ln = lane & 15, lk = lane >> 4;
    if (J >= nblk)
        return;
    // Right-looking over the block column: as soon as W_KJ is known, every later row's sum
    // T_I = sum_K L_IK W_KJ takes its term — the same terms in the same order as a row-by-row
    // (left-looking) evaluation, so the values are bit for bit the same, but only the update of the
    // NEXT row and its product with V are on the dependent chain; the updates of the rows after it are
    // independent MFMAs issued in between (a dependent v_mfma_f64_16x16x4 costs ~130 cycles, an
    // independent one 64).  Measured: 10.0 k instead of 11.1 k cycles for the widest column (the rest
    // is LDS operand loads and the store drain, not the MFMA chain).
    const int nb = min(6, nblk - J); // blocks in this column, J included
    double4_t T[6];
#pragma unroll
    for (int d = 0; d < 6; d++)
        T[d] = double4_t{0, 0, 0, 0};
    double4_t Wc;
#pragma unroll
    for (int q = 0; q < 4; q++)
        Wc[q] = Vs[J * 272 + (lk + 4 * q) * 17 + ln];
#pragma unroll
    for (int q = 0; q < 4; q++)
        Wg[(long)(16 * J + ln) * ncp + 16 * J + lk + 4 * q] = Wc[q];
#pragma unroll
    for (int d = 0; d < 5; d++)
    {
        if (d + 1 < nb) // wave-uniform
        {
            // L_{J+i, J+d} for the rows still open (A operand: [m = ln][k = lk + 4kk]) and V of the next row
            double a[6][4], v[4];
#pragma unroll
            for (int i = d + 1; i < 6; i++)
#pragma unroll
                for (int kk = 0; kk < 4; kk++)
                    a[i][kk] = (i < nb) ? Ls[(16 * (J + d) + lk + 4 * kk) * LLD + 16 * (J + i) + ln] : 0.0;
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
                v[kk] = Vs[(J + d + 1) * 272 + ln * 17 + lk + 4 * kk];
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
            {
                T[d + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[d + 1][kk], Wc[kk], T[d + 1], 0, 0, 0);
#pragma unroll
                for (int i = d + 2; i < 6; i++)
                    if (i < nb)
                        T[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][kk], Wc[kk], T[i], 0, 0, 0);
            }
            double4_t acc = {0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[kk], T[d + 1][kk], acc, 0, 0, 0);
            Wc = -acc;
#pragma unroll
            for (int q = 0; q < 4; q++)
                Wg[(long)(16 * J + ln) * ncp + 16 * (J + d + 1) + lk + 4 * q] = Wc[q];
        }
    }
}

// deal the 16-column blocks of X (block cb costs cb+1 K-blocks) to the 4 wave groups: largest
// first, each to the least loaded group; a group gets at most two blocks / 6 K-blocks.
// Tabulated (nblk <= 6): nibbles of `tab`, low = first block, high = second (15 = none).
__device__ __forceinline__ void deal_col_blocks(int nblk, int g, int& cbA, int& cbB)
{
    // nblk: 1 -> {0},{},{},{}  2 -> {1},{0},{},{}  3 -> {2},{1},{0},{}  4 -> {3},{2},{1},{0}
    //       5 -> {4},{3},{2},{1,0}  6 -> {5},{4},{3,0},{2,1}
    const unsigned tabs[7] = {0xFFFFFFFFu, 0xFFFFFFF0u, 0xFFFFF0F1u, 0xFFF0F1F2u,
                              0xF0F1F2F3u, 0x01F2F3F4u, 0x1203F4F5u};
    const unsigned e = (tabs[nblk] >> (8 * g)) & 0xFFu;
    const int lo = e & 15, hi = e >> 4;
    cbA = lo == 15 ? -1 : lo;
    cbB = hi == 15 ? -1 : hi;
}

// rows [row0, row0+nrows) (absolute scalar rows, nrows <= 64) of F21:  X = B W^T  (X L11^T = B)
// as a GEMM on the f64 matrix cores.  The B tile is staged in LDS k-major (Bt[k*PST + r]); wave
// w owns row group w&3 and one or two 16-column blocks cb of X (dealt so that every wave has at
// most 6 K-blocks of work: X block cb needs K-blocks 0..cb because W is lower triangular).
// A-operand = W[16cb+m][k] straight from global (coalesced, L2-resident, shared by all tiles
// of the front), all loads issued before the barrier.  D[m][n] = X[row n][col m]: the 16 lanes
// l&15 store consecutive rows of one column.
constexpr int PSTB = 80;
__device__ __forceinline__ void dev_trsm_w(double* __restrict__ F, long ld, int nc, long row0, int nrows,
                                           const double* __restrict__ Wg, double* __restrict__ Bt,
                                           double* __restrict__ junk, double* __restrict__ L21c = nullptr,
                                           long ld2 = 0)
{
    // the wave index as a scalar (roles, K-block counts and column indices then live in SGPRs and the
    // branches on them are scalar ones); rows / columns past the end are clamped for the loads and sent
    // to a private sink slot for the stores: no exec-mask region per access
    const int ncp = pad16(nc), nblk = ncp >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, ln = lane & 15, lk = lane >> 4;
    const int rg = w & 3, g = w >> 2;
    double* sink = junk + ((blockIdx.x & 63) << 10) + threadIdx.x;
    int cbA, cbB;
    deal_col_blocks(nblk, g, cbA, cbB);
    const int nA = cbA + 1, nU = nA + cbB + 1; // K-blocks of the first block / in total (<= 6)
    double a[6][4];
#pragma unroll
    for (int u = 0; u < 6; u++)
    {
        const int cb = u < nA ? cbA : cbB, kb = u < nA ? u : u - nA;
        if (u < nU) // scalar
        {
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
                a[u][kk] = Wg[(long)(16 * kb + lk + 4 * kk) * ncp + 16 * cb + ln];
        }
        else
        {
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
                a[u][kk] = 0.0;
        }
    }
    { // stage the B tile: r = lane, 6 columns per thread (k = w + 16u), loads first
        const int r = lane;
        const bool rok = r < nrows;
        const long rc = row0 + min(r, nrows - 1);
        double v[6];
#pragma unroll
        for (int u = 0; u < 6; u++)
        {
            const int k = w + 16 * u; // scalar
            const double x = F[(long)min(k, nc - 1) * ld + rc];
            v[u] = (k < nc && rok) ? x : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 6; u++)
        {
            const int k = w + 16 * u;
            if (k < ncp) // scalar
                Bt[k * PSTB + r] = v[u];
        }
    }
    TILE_SYNC();
    stamp(1, 3);
    double4_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 6; u++)
    {
        if (u < nU)
        {
            const int kb = u < nA ? u : u - nA;
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
            {
                const double b = Bt[(16 * kb + lk + 4 * kk) * PSTB + 16 * rg + ln];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][kk], b, acc, 0, 0, 0);
            }
            if (u == nA - 1 || u == nU - 1)
            { // block finished: X[row0 + 16rg + ln][16cb + lk + 4q]
                const int cb = u < nA ? cbA : cbB;
                const int r = 16 * rg + ln;
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    const int c = 16 * cb + lk + 4 * q;
                    const bool ok = c < nc && r < nrows;
                    double* d0 = ok ? F + ((long)c * ld + row0 + r) : sink;
                    *d0 = acc[q];
                    if (L21c) // the compact copy the backward substitution reads (row index below the pivots); scalar
                    {
                        double* d1 = ok ? L21c + ((long)c * ld2 + (row0 - nc) + r) : sink;
                        *d1 = acc[q];
                    }
                }
                acc = double4_t{0, 0, 0, 0};
            }
        }
    }
    stamp(1, 4);
    TILE_SYNC();
}

// U(ti,tj) -= L21(ti rows) L21(tj rows)^T, 64x64 tiles on the f64 matrix cores.  The whole
// 1024-thread workgroup (16 waves) works on ONE tile: wave w owns the 16x16 sub-tile
// (columns 16*(w&3).., rows 16*(w>>2)..), i.e. one MFMA accumulator chain per wave.  A single
// wave issues roughly one instruction per 8 cycles and runs its MFMAs back to back at 64 cycles
// each, while four waves per SIMD interleave freely — so the tile is spread over as many waves
// as there are sub-tiles.  The two 64-row panels go through LDS for the whole pivot width
// KC (k-major, stride 80 doubles: the two 16-lane halves of a ds_read_b64 then hit different
// banks): one global round trip and one barrier per tile.
// MFMA operand map (v_mfma_f64_16x16x4_f64): lane l supplies A[m = l&15][k = l>>4] and
// B[k = l>>4][n = l&15]; result reg q holds D[m = (l>>4) + 4q][n = l&15].  With m = U column
// and n = U row the 16 lanes l&15 hit consecutive rows of one column: 128-B segments.
constexpr int PST = 80;  // LDS panel stride (doubles) per k
constexpr int KC_SYRK = 96; // K extent staged per barrier pair (= the widest pivot block)
constexpr int syrk_lds() { return 2 * KC_SYRK * PST; } // doubles
// fused trsm+syrk tile kernel: two panels at stride TPST plus the lower block triangle of W
constexpr int TPST = 72;
constexpr int trsyrk_lds() { return 2 * KC_SYRK * TPST + 21 * 256; } // doubles

template <int KC = KC_SYRK>
__device__ __forceinline__ void dev_syrk_tiles(double* __restrict__ F, long ld, int ncs, int nt, int nrs,
                               int first_tile, int ntiles_total, int ntj, double* __restrict__ lds,
                               double* __restrict__ junk)
{
    double* sink = junk + ((blockIdx.x & 63) << 10) + threadIdx.x; // private slot of this lane
    // tile index -> (ti, tj): column-major over the lower triangle of tiles, nti = ceil(nt/64)
    const int nti = (nt + 63) >> 6;
    const int tt = threadIdx.x;
    const int lane = tt & 63, w = tt >> 6;
    const int wc = w & 3, wr = w >> 2;
    const int ln = lane & 15, lk = lane >> 4;
    const double* L21 = F + ncs; // element (row i, col k) = L21[k*ld + i]
    double* Pi = lds;
    double* Pj = Pi + KC * PST;
    double* U = F + (long)ncs * ld + ncs;
    for (int tile = first_tile; tile < ntiles_total; tile++)
    {
        int ti = 0, tj = 0;
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
        double4_t acc = {0, 0, 0, 0};
        double uold[4];
        stamp(4, 2);
        for (int kc = 0; kc < ncs; kc += KC)
        {
            const int kn = min(KC, ncs - kc);
            { // all global loads of a thread are issued before the first LDS store
                const int r = tt & 63, kq = tt >> 6;
                const int gi = 64 * ti + r, gj = 64 * tj + r;
                double vi[KC / 16], vj[KC / 16];
#pragma unroll
                for (int u = 0; u < KC / 16; u++)
                {
                    const int k = kq + 16 * u;
                    vi[u] = (k < kn && gi < nt) ? L21[(long)(kc + k) * ld + gi] : 0.0;
                    vj[u] = (!diag && k < kn && gj < nt) ? L21[(long)(kc + k) * ld + gj] : 0.0;
                }
                if (kc == 0)
                { // the U entries this lane updates travel with the first chunk
#pragma unroll
                    for (int q = 0; q < 4; q++)
                    { // masked lanes read their private sink slot: no branch
                        const int i = 64 * ti + 16 * wr + ln, j = 64 * tj + 16 * wc + lk + 4 * q;
                        const bool ok = i < nt && j < nrs && i >= j;
                        const double* src = ok ? U + ((long)j * ld + i) : sink;
                        uold[q] = *src;
                    }
                }
#pragma unroll
                for (int u = 0; u < KC / 16; u++)
                {
                    const int k = kq + 16 * u;
                    if (k < kn)
                    {
                        Pi[k * PST + r] = vi[u];
                        if (!diag)
                            Pj[k * PST + r] = vj[u];
                    }
                }
            }
            TILE_SYNC();
            {
                const double* Pa = diag ? Pi : Pj;
                for (int k0 = 0; k0 < kn; k0 += 4)
                {
                    const int k = min(k0 + lk, kn - 1); // clamped: no branch around the LDS reads
                    const bool kok = k0 + lk < kn;
                    const double a = Pa[k * PST + 16 * wc + ln];
                    const double b = Pi[k * PST + 16 * wr + ln];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kok ? a : 0.0, b, acc, 0, 0, 0);
                }
            }
            TILE_SYNC();
        }
        stamp(4, 3);
#pragma unroll
        for (int q = 0; q < 4; q++)
        { // branch-free: masked lanes store into their sink slot
            const int i = 64 * ti + 16 * wr + ln, j = 64 * tj + 16 * wc + lk + 4 * q;
            const bool ok = i < nt && j < nrs && i >= j;
            double* dst = ok ? U + ((long)j * ld + i) : sink;
            *dst = uold[q] - acc[q];
        }
    }
}

// Fused trsm + syrk for ONE 64x64 tile (ti, tj) of the update matrix of a front (upper stages):
//   X_i = B_i W^T, X_j = B_j W^T   (B = rows of F21 of tile rows ti / tj, 64 x ncs each)
//   U(ti,tj) -= X_i X_j^T
// Every tile recomputes the two L21 row tiles it needs from B and W instead of waiting for a
// separate trsm kernel: the redundant MFMAs are cheap next to a kernel boundary (launch gap +
// prologue + the L21 round trip through memory), and the whole tile is still one global load
// round trip.  B stays untouched in the front (other tiles read it concurrently); the tile
// with tj == ti (or tj < 0: a tile row that has no diagonal tile) stores X_i to the compact
// L21 buffer that the backward substitution reads.
// LDS: Pi | Pj, k-major panels (stride PST) that first hold B and then, in place, X.
// qmask (look-ahead schedule): the leading qmask x qmask block of U and the first qmask rows of
// L21 belong to the front's lead workgroup (LEAD = true, called as tile (0,0) with nt = nrs = q, or
// as tile (1,0) when 64 < q <= 96: that one then also does the two diagonal tiles from the X panels
// it has in LDS anyway and stores the L21 rows of both) and are not stored by the other tiles.
template <bool LEAD>
__device__ __forceinline__ void dev_trsyrk_tile(double* __restrict__ F, long ld, int ncs, int nt, int nrs,
                                                int ti, int tj, const double* __restrict__ Wg,
                                                double* __restrict__ L21, long ld2,
                                                double* __restrict__ lds, double* __restrict__ junk,
                                                int qmask)
{
    constexpr int KC = KC_SYRK;
    constexpr int PST = TPST; // panel stride of this kernel (leaves LDS room for W)
    double* sink = junk + ((blockIdx.x & 63) << 10) + threadIdx.x; // private slot of this lane
    const int ncp = pad16(ncs), nblk = ncp >> 4;
    const int tt = threadIdx.x, lane = tt & 63, ln = lane & 15, lk = lane >> 4;
    // the wave index as a scalar: everything derived from it (roles, block assignment, K-block
    // counts, column indices) then lives in SGPRs and branches on it are scalar branches — an
    // exec-mask region per conditional load costs ~14 instructions, the load itself one
    const int w = __builtin_amdgcn_readfirstlane(tt >> 6);
    const int rg = w & 3, g = w >> 2; // trsm role: row group, column-block group
    const int wc = w & 3, wr = w >> 2; // syrk role: column / row sub-tile
    const bool solo = tj < 0, diag = tj == ti, two = !solo && !diag;
    double* Pi = lds;
    double* Pj = lds + KC * PST;
    double* Wl = lds + 2 * KC * PST; // lower block triangle of W, block (I,K) at (I(I+1)/2 + K)*256, [k][m]
    double* U = F + (long)ncs * ld + ncs;
    int cbA, cbB;
    deal_col_blocks(nblk, g, cbA, cbB);
    const int nA = cbA + 1, nU = nA + cbB + 1;
    stamp(4, 1);
    // ---- all global loads first: W (once per workgroup, into LDS: every wave re-loading its own
    // operands costs 4x the cache-line traffic and the load phase is bound by the per-CU line
    // rate), the B tile(s), the U entries to update
    const int nwblk = nblk * (nblk + 1) / 2;
    double wv[6];
    int wdst[6];
#pragma unroll
    for (int u = 0; u < 6; u++)
    {
        const int f = tt + u * 1024; // element of the staged triangle: block f>>8, k = (f>>4)&15, m = f&15
        const int blk = min(f >> 8, nwblk - 1);
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= blk)
            I++;
        const int K = blk - I * (I + 1) / 2;
        const int k = (f >> 4) & 15, m = f & 15;
        wdst[u] = (f >> 8) < nwblk ? f : -1;
        wv[u] = ldg32(Wg, (unsigned)(16 * K + k) * (unsigned)ncp + (unsigned)(16 * I + m));
    }
    {
        const int r = lane, kq = w;
        const int gi = 64 * ti + r, gj = 64 * tj + r;
        const double* B = F + ncs; // (row i, col k) = B[k*ld + i]; 32-bit offsets from the uniform base
        const unsigned uld = (unsigned)ld;
        // rows past the end are clamped (a valid address, no branch) and zeroed afterwards
        const unsigned gic = (unsigned)min(gi, nt - 1), gjc = (unsigned)min(max(gj, 0), nt - 1);
        const bool iok = gi < nt, jok = two && gj < nt;
        double vi[6], vj[6];
#pragma unroll
        for (int u = 0; u < 6; u++)
        {
            const int k = kq + 16 * u; // scalar
            const unsigned kc = (unsigned)min(k, ncs - 1);
            const double bi = ldg32(B, kc * uld + gic);
            const double bj = two ? ldg32(B, kc * uld + gjc) : 0.0; // `two` is uniform
            vi[u] = (k < ncs && iok) ? bi : 0.0;
            vj[u] = (k < ncs && jok) ? bj : 0.0;
        }
        double uold[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
        { // masked lanes read their private sink slot: no branch
            const int i = 64 * ti + 16 * wr + ln, j = 64 * tj + 16 * wc + lk + 4 * q;
            const bool ok = !solo && i < nt && j < nrs && i >= j;
            const double* src = ok ? U + ((long)j * ld + i) : sink;
            uold[q] = *src;
        }
        double uold2[2][4]; // LEAD, two panels: the diagonal tiles (tj,tj) and (ti,ti)
        if (LEAD && two)
        {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    const int tb = h == 0 ? tj : ti;
                    const int i = 64 * tb + 16 * wr + ln, j = 64 * tb + 16 * wc + lk + 4 * q;
                    const bool ok = i < nt && j < nrs && i >= j;
                    const double* src = ok ? U + ((long)j * ld + i) : sink;
                    uold2[h][q] = *src;
                }
        }
#pragma unroll
        for (int u = 0; u < 6; u++)
        {
            const int k = kq + 16 * u;
            if (k < ncp)
            {
                Pi[k * PST + r] = vi[u];
                if (two)
                    Pj[k * PST + r] = vj[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 6; u++)
            if (wdst[u] >= 0)
                Wl[wdst[u]] = wv[u];
        stamp(4, 2);
        TILE_SYNC();
        stamp(4, 3);
        // ---- X = B W^T for this wave's row group and column blocks (both panels)
        double4_t xi[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, xj[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        {
            double4_t ai = {0, 0, 0, 0}, aj = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 6; u++)
            {
                if (u < nU)
                {
                    const int cb = u < nA ? cbA : cbB, kb = u < nA ? u : u - nA;
                    const double* Wb = Wl + (cb * (cb + 1) / 2 + kb) * 256 + lk * 16 + ln; // W[16cb+m][16kb+k]
#pragma unroll
                    for (int kk = 0; kk < 4; kk++)
                    {
                        const int ko = (16 * kb + lk + 4 * kk) * PST + 16 * rg + ln;
                        const double av = Wb[64 * kk];
                        ai = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Pi[ko], ai, 0, 0, 0);
                        if (two)
                            aj = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Pj[ko], aj, 0, 0, 0);
                    }
                    if (u == nA - 1)
                        xi[0] = ai, xj[0] = aj, ai = double4_t{0, 0, 0, 0}, aj = double4_t{0, 0, 0, 0};
                    else if (u == nU - 1)
                        xi[1] = ai, xj[1] = aj;
                }
            }
        }
        stamp(4, 4);
        TILE_SYNC(); // every wave has read B: the panels may be overwritten with X
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const int cb = h == 0 ? cbA : cbB;
            if (cb >= 0)
            {
                const int row = 64 * ti + 16 * rg + ln;
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    const int c = 16 * cb + lk + 4 * q;
                    Pi[c * PST + 16 * rg + ln] = xi[h][q];
                    if (two)
                        Pj[c * PST + 16 * rg + ln] = xj[h][q];
                    if ((!two || LEAD) && c < ncs && row < nt && row >= qmask) // diag or solo: this tile owns L21 rows ti
                        L21[(long)c * ld2 + row] = xi[h][q];
                    if (LEAD && two && c < ncs) // rows tj (all below nt: ti > tj)
                        L21[(long)c * ld2 + 64 * tj + 16 * rg + ln] = xj[h][q];
                }
            }
        }
        TILE_SYNC();
        stamp(4, 5);
        if (!solo)
        { // ---- U(ti,tj) -= X_i X_j^T, one 16x16 sub-tile per wave
            const double* Pa = diag ? Pi : Pj;
            double4_t acc = {0, 0, 0, 0};
            for (int k0 = 0; k0 < ncs; k0 += 4)
            {
                const int k = min(k0 + lk, ncs - 1); // clamped: no branch around the LDS reads
                const bool kok = k0 + lk < ncs;
                const double av = Pa[k * PST + 16 * wc + ln];
                const double bv = Pi[k * PST + 16 * wr + ln];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kok ? av : 0.0, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
            { // branch-free: masked lanes store into their sink slot
                const int i = 64 * ti + 16 * wr + ln, j = 64 * tj + 16 * wc + lk + 4 * q;
                const bool ok = i < nt && j < nrs && i >= j && !(i < qmask && j < qmask);
                double* dst = ok ? U + ((long)j * ld + i) : sink;
                *dst = uold[q] - acc[q];
            }
            if (LEAD && two)
            { // the diagonal tiles of both panels: U(tb,tb) -= X_tb X_tb^T
#pragma unroll
                for (int h = 0; h < 2; h++)
                {
                    const int tb = h == 0 ? tj : ti;
                    const double* Pd = h == 0 ? Pj : Pi;
                    if (wr >= wc && 64 * tb + 16 * wr < nt) // else: above the diagonal or past the end (uniform per wave)
                    {
                        double4_t a2 = {0, 0, 0, 0};
                        for (int k0 = 0; k0 < ncs; k0 += 4)
                        {
                            const int k = min(k0 + lk, ncs - 1);
                            const bool kok = k0 + lk < ncs;
                            const double av = Pd[k * PST + 16 * wc + ln];
                            const double bv = Pd[k * PST + 16 * wr + ln];
                            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(kok ? av : 0.0, bv, a2, 0, 0, 0);
                        }
#pragma unroll
                        for (int q = 0; q < 4; q++)
                        {
                            const int i = 64 * tb + 16 * wr + ln, j = 64 * tb + 16 * wc + lk + 4 * q;
                            const bool ok = i < nt && j < nrs && i >= j;
                            double* dst = ok ? U + ((long)j * ld + i) : sink;
                            *dst = uold2[h][q] - a2[q];
                        }
                    }
                }
            }
        }
    }
}

// The same for ONE 32x32 tile: used on the levels with few fronts (the separator chains near the
// root), where a launch has far fewer 64x64 tiles than the chip has CUs and its duration is the
// duration of one tile — which is bound by the f64 matrix pipe of one CU (trsm of two 64-row
// panels: 672 MFMAs, syrk: 384; 64 cycles each, four SIMDs).  A 32x32 tile needs 336 + 96 and
// there are four times as many of them to spread over the idle CUs.
//   trsm roles  : wave w -> row group w & 1 (16 rows), column block of X dealt by (w >> 1) so that
//                 the waves sharing a SIMD (w mod 4) carry 11 / 10 K-blocks in total
//   syrk roles  : waves 0..3 (one per SIMD), one 16x16 sub-tile each over the whole pivot width
constexpr int TPST32 = 48; // panel stride: the two 16-lane halves of a ds_read_b64 hit different banks
__device__ __forceinline__ void dev_trsyrk_tile32(double* __restrict__ F, long ld, int ncs, int nt, int nrs,
                                                  int ti, int tj, const double* __restrict__ Wg,
                                                  double* __restrict__ L21, long ld2,
                                                  double* __restrict__ lds, double* __restrict__ junk,
                                                  int qmask)
{
    constexpr int KC = KC_SYRK;
    constexpr int PST = TPST32;
    double* sink = junk + ((blockIdx.x & 63) << 10) + threadIdx.x; // private slot of this lane
    const int ncp = pad16(ncs), nblk = ncp >> 4;
    const int tt = threadIdx.x, lane = tt & 63, ln = lane & 15, lk = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tt >> 6);
    const int rg = w & 1, g = w >> 1;
    const bool solo = tj < 0, diag = tj == ti, two = !solo && !diag;
    double* Pi = lds;
    double* Pj = lds + KC * PST;
    double* Wl = lds + 2 * KC * PST; // lower block triangle of W, block (I,K) at (I(I+1)/2 + K)*256, [k][m]
    double* U = F + (long)ncs * ld + ncs;
    // column block of this wave: the blocks in the order of their cost (block cb needs cb+1
    // K-blocks) go to the groups 0,1,3,2,4,5,7,6: even and odd groups — which sit on different
    // SIMD pairs — then carry 6+3+2 and 5+4+1 K-blocks
    const int pos = (0x67542310 >> (4 * g)) & 15; // position of group g in that order (nibble g)
    const int cb = nblk - 1 - pos, nK = cb + 1;         // cb < 0: no block
    // ---- all global loads first: W into LDS (once per workgroup), the B tile(s), the U entries
    const int nwblk = nblk * (nblk + 1) / 2;
    double wv[6];
    int wdst[6];
#pragma unroll
    for (int u = 0; u < 6; u++)
    {
        const int f = tt + u * 1024; // element of the staged triangle: block f>>8, k = (f>>4)&15, m = f&15
        const int blk = min(f >> 8, nwblk - 1);
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= blk)
            I++;
        const int K = blk - I * (I + 1) / 2;
        const int k = (f >> 4) & 15, m = f & 15;
        wdst[u] = (f >> 8) < nwblk ? f : -1;
        wv[u] = ldg32(Wg, (unsigned)(16 * K + k) * (unsigned)ncp + (unsigned)(16 * I + m));
    }
    const int r = lane & 31, kq = 2 * w + (lane >> 5); // B staging: 32 rows x 32 columns per round
    const int gi = 32 * ti + r, gj = 32 * tj + r;
    const double* B = F + ncs; // (row i, col k) = B[k*ld + i]
    const unsigned uld = (unsigned)ld;
    const unsigned gic = (unsigned)min(gi, nt - 1), gjc = (unsigned)min(max(gj, 0), nt - 1);
    const bool iok = gi < nt, jok = two && gj < nt;
    double vi[3], vj[3];
#pragma unroll
    for (int u = 0; u < 3; u++)
    {
        const int k = kq + 32 * u;
        const unsigned kc = (unsigned)min(k, ncs - 1);
        const double bi = ldg32(B, kc * uld + gic);
        const double bj = two ? ldg32(B, kc * uld + gjc) : 0.0;
        vi[u] = (k < ncs && iok) ? bi : 0.0;
        vj[u] = (k < ncs && jok) ? bj : 0.0;
    }
    const int wc = w & 1, wr = (w >> 1) & 1; // syrk role of waves 0..3
    double uold[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const int i = 32 * ti + 16 * wr + ln, j = 32 * tj + 16 * wc + lk + 4 * q;
        const bool ok = !solo && w < 4 && i < nt && j < nrs && i >= j;
        const double* src = ok ? U + ((long)j * ld + i) : sink;
        uold[q] = *src;
    }
#pragma unroll
    for (int u = 0; u < 3; u++)
    {
        const int k = kq + 32 * u;
        if (k < ncp)
        {
            Pi[k * PST + r] = vi[u];
            if (two)
                Pj[k * PST + r] = vj[u];
        }
    }
#pragma unroll
    for (int u = 0; u < 6; u++)
        if (wdst[u] >= 0)
            Wl[wdst[u]] = wv[u];
    TILE_SYNC();
    // ---- X = B W^T for this wave's row group and column block (both panels)
    double4_t xi = {0, 0, 0, 0}, xj = {0, 0, 0, 0};
    if (cb >= 0)
    {
        for (int kb = 0; kb < nK; kb++)
        {
            const double* Wb = Wl + (cb * (cb + 1) / 2 + kb) * 256 + lk * 16 + ln; // W[16cb+m][16kb+k]
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
            {
                const int ko = (16 * kb + lk + 4 * kk) * PST + 16 * rg + ln;
                const double av = Wb[64 * kk];
                xi = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Pi[ko], xi, 0, 0, 0);
                if (two)
                    xj = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Pj[ko], xj, 0, 0, 0);
            }
        }
    }
    TILE_SYNC(); // every wave has read B: the panels may be overwritten with X
    if (cb >= 0)
    {
        const int row = 32 * ti + 16 * rg + ln;
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int c = 16 * cb + lk + 4 * q;
            Pi[c * PST + 16 * rg + ln] = xi[q];
            if (two)
                Pj[c * PST + 16 * rg + ln] = xj[q];
            if (!two && c < ncs && row < nt && row >= qmask) // diag or solo: this tile owns L21 rows ti
                L21[(long)c * ld2 + row] = xi[q];
        }
    }
    TILE_SYNC();
    if (!solo && w < 4)
    { // ---- U(ti,tj) -= X_i X_j^T, one 16x16 sub-tile per wave
        const double* Pa = diag ? Pi : Pj;
        double4_t acc = {0, 0, 0, 0};
        for (int k0 = 0; k0 < ncs; k0 += 4)
        {
            const int k = min(k0 + lk, ncs - 1); // clamped: no branch around the LDS reads
            const bool kok = k0 + lk < ncs;
            const double av = Pa[k * PST + 16 * wc + ln];
            const double bv = Pi[k * PST + 16 * wr + ln];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kok ? av : 0.0, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int i = 32 * ti + 16 * wr + ln, j = 32 * tj + 16 * wc + lk + 4 * q;
            const bool ok = i < nt && j < nrs && i >= j && !(i < qmask && j < qmask);
            double* dst = ok ? U + ((long)j * ld + i) : sink;
            *dst = uold[q] - acc[q];
        }
    }
}

// backward substitution of one front: x_J = W^T (y_J - L21^T x_R), W = L11^-1 from dev_winv.
// Both halves are matrix-vector products: 16 lanes per pivot column, all loads of a lane in
// flight together, a 16-lane butterfly at the end.  No serial triangular solve.
// what the backward substitution needs to know about a front.  A single-front task (every task of
// an upper stage) finds it in ONE 64-byte record indexed by the task (CholPlanDev::tmeta) instead of
// walking task -> front -> eight per-front arrays: two dependent global round trips less per level.
struct FrontView
{
    int ncb, nb, col0, bw_np, rows_ptr;
    long off, ldf, woff, l21off;
};
__device__ __forceinline__ FrontView front_view(const CholPlanDev& p, int f)
{
    return FrontView{p.ncb[f], p.nb[f], p.col0[f], p.bw_np[f], p.rows_ptr[f], p.off[f], p.ldf[f], p.woff[f], p.l21off[f]};
}
__device__ __forceinline__ FrontView front_view(const int32_t* __restrict__ tm)
{
    const long* t64 = reinterpret_cast<const long*>(tm + 8);
    return FrontView{tm[2], tm[3], tm[4], tm[5], tm[6], t64[0], t64[1], t64[2], t64[3]};
}

__device__ __forceinline__ void dev_backward(const CholPlanDev& p, const double* __restrict__ fronts,
                             const FrontView fv, double* __restrict__ lds, double* __restrict__ xnew,
                             double* __restrict__ xout, const int32_t* __restrict__ rows16 = nullptr)
{
    const int ncb = fv.ncb, nb = fv.nb;
    const long ld = fv.ldf;
    const int ncs = 6 * ncb, nrs = 6 * (nb - ncb), ncp = pad16(ncs);
    // L21 and the forward-solved rhs row: compact buffer (upper stages) or the front itself
    const long l21o = fv.l21off;
    const double* L = l21o >= 0 ? p.l21 + l21o : fronts + fv.off + ncs;
    const long ldl = l21o >= 0 ? nrs + 1 : ld;
    const double* Wg = p.winv + fv.woff;
    const int c0 = fv.col0;
    double* vs = lds;      // ncp
    double* xr = lds + ncp; // nrs
    const int32_t* rows = p.rows + fv.rows_ptr;
    stamp(3, 1);
    stamp_value(3, 6, 1000L * ncs + nrs);
    // rows of the mat-vec still to do here: all of them, or — when the ancestor part was done one
    // launch ahead (k_backward_stage, extra workgroups; the partial result waits in xnew at this
    // front's own positions) — only the leading rows that belong to the parent
    const int npb = fv.bw_np;
    const int nr_here = npb >= 0 ? 6 * npb : nrs;
    // 8 lanes per pivot column: the 128 lane groups of the 1024-thread workgroup cover all (<= 96)
    // columns at once, so every thread knows its column now and ALL its global loads — x of the
    // boundary rows, its slice of the first 96 rows of L21, y, its slice of W, the permutation — are
    // issued together and waited for once: the kernel is a chain of global round trips (~2 k cycles
    // each: L21 and W were written by other XCDs), not of arithmetic
    const int g = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    const bool colok = g < ncs;
    const int j = min(g, ncs - 1);
    const double* col = L + (long)j * ldl;
    const double* wcol = Wg + (long)j * ncp;
    const int jb = j / 6, comp = j - 6 * jb;
    double xg = 0.0; // this thread's entry of x_R (nr_here <= blockDim on every front of an upper stage)
    const int ig = threadIdx.x;
    // (rows16: the first 16 boundary block rows ride in the task record — with the ancestor part done ahead
    // these are all the rows there are, and the gather starts one round trip earlier)
    const int32_t* rws = (rows16 && nr_here <= 96) ? rows16 : rows;
    if (ig < nr_here)
        xg = xnew[6L * rws[ig / 6] + (ig % 6)];
    // 16-byte loads: the 8 lanes of a column read one full 128-byte line per instruction (rows 2 l8 + 16 u and the
    // one after it) — half the load instructions and half the cache-line requests of 8-byte loads at stride 8.
    // (col + i is only 8-byte aligned when the column length is odd: global loads take that; index nrs — the
    // rhs entry — exists, so the pair (nrs - 1, nrs) is in bounds)
    double a[12], wv[12];
#pragma unroll
    for (int u = 0; u < 6; u++)
    {
        const double2 v = *reinterpret_cast<const double2*>(col + min(2 * l8 + 16 * u, max(nrs - 1, 0)));
        a[2 * u] = v.x, a[2 * u + 1] = v.y;
    }
    const double y = npb >= 0 ? xnew[6L * c0 + j] : col[nrs];
    const int k0 = (j & ~15) + 2 * l8;
#pragma unroll
    for (int u = 0; u < 6; u++)
    { // (ncp is a multiple of 16 and k0 + 16 u is even: aligned, and k0 + 16 u + 1 < ncp whenever k0 + 16 u < ncp)
        const double2 v = *reinterpret_cast<const double2*>(wcol + min(k0 + 16 * u, ncp - 2));
        wv[2 * u] = v.x, wv[2 * u + 1] = v.y;
    }
    const int pj = p.perm[c0 + jb];
    if (ig < nr_here)
        xr[ig] = xg;
    for (int i = ig + blockDim.x; i < nr_here; i += blockDim.x) // (subtree-stage fronts with > 1024 rows)
    {
        const int ib = i / 6;
        xr[i] = xnew[6L * rows[ib] + (i - 6 * ib)];
    }
    for (int jj = ncs + threadIdx.x; jj < ncp; jj += blockDim.x)
        vs[jj] = 0.0;
    TILE_SYNC();
    stamp(3, 2);
    // v_j = y_j - sum_i L21[i,j] x_R[i]
    {
        double s = 0;
#pragma unroll
        for (int u = 0; u < 12; u++)
        {
            const int i = 2 * l8 + 16 * (u >> 1) + (u & 1);
            s += (i < nr_here) ? a[u] * xr[i] : 0.0;
        }
        for (int i0 = 96; i0 < nr_here; i0 += 96)
        { // rows beyond the first 96 (a front that does its whole mat-vec itself)
            double b[12];
#pragma unroll
            for (int u = 0; u < 6; u++)
            {
                const double2 v = *reinterpret_cast<const double2*>(col + min(i0 + 2 * l8 + 16 * u, max(nrs - 1, 0)));
                b[2 * u] = v.x, b[2 * u + 1] = v.y;
            }
#pragma unroll
            for (int u = 0; u < 12; u++)
            {
                const int i = i0 + 2 * l8 + 16 * (u >> 1) + (u & 1);
                s += (i < nr_here) ? b[u] * xr[i] : 0.0;
            }
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1)
            s += __shfl_xor(s, off, 8);
        if (l8 == 0 && colok)
            vs[j] = y - s;
    }
    TILE_SYNC();
    stamp(3, 3);
    // x_j = sum_k W[k][j] v_k over k >= 16*floor(j/16) (W lower triangular, zeros above the
    // diagonal inside the diagonal block); Wg is column-major, so the sum runs along a column
    {
        double s = 0;
#pragma unroll
        for (int u = 0; u < 12; u++)
        {
            const int k = k0 + 16 * (u >> 1) + (u & 1);
            s += (k < ncp) ? wv[u] * vs[k] : 0.0;
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1)
            s += __shfl_xor(s, off, 8);
        if (l8 == 0 && colok)
        {
            xnew[6L * (c0 + jb) + comp] = s;
            xout[6L * pj + comp] = s;
        }
    }
    stamp(3, 4);
    __threadfence_block();
    TILE_SYNC();
}

constexpr int BIG = 1024; // workgroup size of the latency-critical single-front kernels

// CUGO_KERNEL_ACQUIRE=1: every kernel of the factorisation starts with an agent-scope acquire fence of its own
// (buffer_inv sc1: the CU's vector cache and the non-coherent lines of its L2 are dropped) on top of what the
// dispatch does between two kernels of a stream — see DESIGN.md section 2 (the rare run-to-run deviation reads
// like a consumer kernel seeing the PREVIOUS factorisation's value of something its producer just rewrote)
__device__ __forceinline__ void kernel_acquire(const CholPlanDev& p)
{
#ifdef CUGO_DEBUG_HOOKS
    if (p.kernel_acquire & 1)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    (void)p;
#endif
}
// CUGO_KERNEL_ACQUIRE=2 (3: both): every wave ends with an agent-scope release fence of its own (buffer_wbl2 sc1 and a
// wait for all its stores) — the writer's side of the same question
__device__ __forceinline__ void kernel_release(const CholPlanDev& p)
{
#ifdef CUGO_DEBUG_HOOKS
    if (p.kernel_acquire & 2)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else if (p.kernel_acquire & 4) // 4: every wave only waits until its own stores are acknowledged before it ends
        __builtin_amdgcn_s_waitcnt(0); // (vmcnt counts stores too on gfx9; a workgroup-scope fence emits nothing here)
#else
    (void)p;
#endif
}

// CUGO_DEBUG_ZERO_LDS=1 / 2: every kernel of the factorisation first fills its dynamic LDS with zeros / NaNs (the
// launch functions put the size into their copy of the plan) — nothing may depend on what the kernel that ran on
// the CU before has left there
__device__ __forceinline__ void dbg_fill_lds(const CholPlanDev& p, double* lds)
{
#ifdef CUGO_DEBUG_HOOKS
    if (p.zero_lds)
    {
        const double v = p.zero_lds == 2 ? __longlong_as_double(0x7FF8000000000000LL) : 0.0;
        for (int i = threadIdx.x; i < p.lds_doubles; i += blockDim.x)
            lds[i] = v;
        __syncthreads();
    }
#else
    (void)p, (void)lds;
#endif
}

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
        const long ld = p.ldf[f];
        const int ncs = 6 * ncb, nrs = 6 * (nb - ncb), nt = nrs + 1;
        double* F = fronts + p.off[f];
        const int ncp = pad16(ncs);
        double* Ls = lds;
        double* dinv = lds + NC_MAX * LLD;
        double* Vs = dinv + NC_MAX;
        double* Bt = Vs; // the trsm tile overlays the diagonal-block inverses (dead once W is built)
        double* Wg = p.winv + p.woff[f];
        if (p.alias_of[f] < 0) // a front stored in its only child's update block needs no extend-add
            dev_extend_add(p, fronts, f, 0, nb, 0);
        dev_potrf_load(F, ld, ncs, Ls, dinv);
        dev_potrf_panels(ncs, Ls, dinv, fail);
        __syncthreads();
        dev_inv_diag16(Ls, dinv, ncp, Vs);
        __syncthreads();
        dev_winv(Ls, ncp, Vs, Wg);
        __threadfence_block();
        __syncthreads();
        for (int r0 = 0; r0 < nt; r0 += TR)
            dev_trsm_w(F, ld, ncs, ncs + r0, min(TR, nt - r0), Wg, Bt, p.junk);
        __threadfence_block();
        __syncthreads();
        const int nti = (nt + 63) / 64, ntj = (nrs + 63) / 64;
        int ntiles = 0;
        for (int tj = 0; tj < ntj; tj++)
            ntiles += nti - tj;
        dev_syrk_tiles(F, ld, ncs, nt, nrs, 0, ntiles, ntj, lds, p.junk);
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------- upper stages ---------
// potrf of the stage's fronts and, in the same launch (blocks >= npotrf), the extend-add of
// everything except the parents' F11 blocks: independent data, no separate extend-add launch
__global__ __launch_bounds__(BIG) void k_up_potrf(CholPlanDev p, double* __restrict__ fronts,
                                                  int task0, int npotrf,
                                                  const int32_t* __restrict__ wl_eap, int neap,
                                                  const int32_t* __restrict__ wl_eab,
                                                  int32_t* __restrict__ fail)
{
#ifdef CUGO_DEBUG_HOOKS
    if (p.dbg_skip_wg == (int)blockIdx.x)
        return; // (fault injection, see DbgSkip)
#endif
    kernel_acquire(p);
    extern __shared__ double lds[];
    if ((int)blockIdx.x >= npotrf)
    { // items of the pivot columns first, then of the boundary columns
        const int b = blockIdx.x - npotrf;
        if (DBG_DELAY(p) == 6)
            dbg_sleep();
        if (DBG_DELAY(p) == 28) // (omission patterns, see dev_potrf16: 28 = the floor of the potrf workgroups alone,
            return;             //  29 = these extend-add workgroups alone)
        const int32_t* it = b < neap ? wl_eap + 3 * b : wl_eab + 3 * (b - neap);
        dev_extend_add(p, fronts, it[0], it[1], it[2], 2); // everything below the parents' F11
        kernel_release(p);
        return;
    }
    stamp(0, 0);
    if (DBG_DELAY(p) == 5)
        dbg_sleep();
    if (DBG_DELAY(p) == 29)
        return;
    dbg_fill_lds(p, lds);
    // the front's scalars from the task's 64-byte record (one scalar load instead of task -> front -> arrays)
    const int32_t* tm = p.tmeta + cugo_k::TMETA * (task0 + blockIdx.x);
    const long* tm64 = reinterpret_cast<const long*>(tm + 8);
    const int f = tm[1];
    const int ncs = 6 * tm[2];
    const long foff = tm64[0], fld = tm64[1], fwoff = tm64[2];
    // the part of the extend-add this factorisation depends on: the children's contributions to
    // F11 (<= 96 x 96), gathered by this workgroup itself; the rest of the pivot columns and the
    // boundary columns are gathered meanwhile by the extra workgroups above
    const int ncp = pad16(ncs);
    double* Ls = lds;
    double* dinv = lds + NC_MAX * LLD;
    double* Vs = dinv + NC_MAX;
    const bool kids = tm[7] != 0; // has children and is not stored inside its only child
    if (kids && !p.ea_lds)
    { // (CUGO_EA_LDS=0: through the front in memory, then loaded like the rest)
        dev_extend_add(p, fronts, f, 0, tm[2], 1);
        __threadfence_block();
        __syncthreads();
    }
    const bool mirror_now = p.panel16 && !kids; // nothing will be added to F11: the diagonal tiles go in symmetric
    if (mirror_now)
        dev_potrf_load<true>(fronts + foff, fld, ncs, Ls, dinv);
    else
        dev_potrf_load(fronts + foff, fld, ncs, Ls, dinv);
    stamp(0, 1);
    if (kids && p.ea_lds)
    { // straight into the LDS copy (Vs: unused until the W phase, serves as the masked lanes' sink).
      // (Measured, not kept: the rel entries and update-block entries of two children fetched into registers
      // before / beside the loads of F11 — every global load of the phase in flight at once, only the LDS adds
      // child after child: 11.57 vs 11.22 ms per step on the kitti_00 shape, 38.1 vs 37.5 ms on the 10k-pose
      // graph.  The phase is not bound by the latency of its loads.)
        // (a second attempt — the NEXT child's rel and update-block entries fetched into registers while the current
        // child's are added, nothing else changed — was slower as well: 11.14 vs 10.97 ms, 38.6 vs 37.9 ms)
        // (a third attempt, round 4: the child links as ONE 128-byte line each — record + the 16 rel entries its leading
        // rows can use — fetched beside F11 and kept in LDS, i.e. one dependent global round trip less per child and
        // none for the first record: 20.2 vs 20.1 us per launch, nothing.  Without this call altogether — omission
        // pattern 30 — the ten levels with children run 3 - 5 us shorter, 44 us per factorisation.)
        if (DBG_DELAY(p) != 30)
            dev_extend_add_lead(p, fronts, tm[16], tm[17], Ls, Vs + threadIdx.x);
    }
    if (p.panel16)
    { // 16-column L D L^T panels, W built behind them (dev_potrf16)
        dev_potrf16(ncp, lds, p.winv + fwoff, fail, mirror_now, DBG_DELAY(p));
        stamp_value(0, 6, ncs);
        stamp(0, 7);
        kernel_release(p);
        return;
    }
    dev_potrf_panels(ncs, Ls, dinv, fail);
    __syncthreads();
    stamp(0, 5);
    // one wave per diagonal block: V_J = inverse of the 16x16 diagonal block
    {
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        if (wv < (ncp >> 4))
            inv_diag16_wave(Ls, dinv, wv, Vs);
    }
    __syncthreads();
    dev_winv(Ls, ncp, Vs, p.winv + fwoff);
    stamp_value(0, 6, ncs);
    stamp(0, 7);
    kernel_release(p);
}

// fused trsm + syrk: one workgroup per item (front, ti, tj), see dev_trsyrk_tile
// A tile / row-tile item with everything its kernel needs about the front in ONE 64-byte record
// (CholPlanDev::fat, parallel to the item list): {front, a, b, 6 ncb, 6 (nb - ncb), -, -, -, off, ldf,
// woff, l21off}.  item -> front -> six per-front arrays was one dependent round trip more at the head
// of every tile kernel.
struct TileItem
{
    int a, b, ncs, nrs;
    long off, ld, woff, l21off;
};
__device__ __forceinline__ TileItem tile_item(const CholPlanDev& p, const int32_t* __restrict__ wl)
{
    const int32_t* ft = p.fat + 16 * ((int)((wl - p.wl_base) / 3) + (int)blockIdx.x);
    const long* f64 = reinterpret_cast<const long*>(ft + 8);
    return TileItem{ft[1], ft[2], ft[3], ft[4], f64[0], f64[1], f64[2], f64[3]};
}

__global__ __launch_bounds__(BIG) void k_up_trsyrk(CholPlanDev p, double* __restrict__ fronts,
                                                   const int32_t* __restrict__ wl)
{
#ifdef CUGO_DEBUG_HOOKS
    if (p.dbg_skip_wg == (int)blockIdx.x)
        return; // (fault injection, see DbgSkip)
#endif
    kernel_acquire(p);
    extern __shared__ double lds[];
    dbg_fill_lds(p, lds);
    stamp(4, 0);
    const TileItem t = tile_item(p, wl);
    dev_trsyrk_tile<false>(fronts + t.off, t.ld, t.ncs, t.nrs + 1, t.nrs, t.a, t.b, p.winv + t.woff,
                           p.l21 + t.l21off, t.nrs + 1, lds, p.junk, 0);
    stamp(4, 7);
    stamp_value(4, 6, 1000000L * (t.a * 10 + t.b + 1) + 1000L * t.ncs + t.nrs);
    kernel_release(p);
}

__global__ __launch_bounds__(BIG) void k_up_trsyrk32(CholPlanDev p, double* __restrict__ fronts,
                                                     const int32_t* __restrict__ wl)
{
#ifdef CUGO_DEBUG_HOOKS
    if (p.dbg_skip_wg == (int)blockIdx.x)
        return; // (fault injection, see DbgSkip)
#endif
    kernel_acquire(p);
    extern __shared__ double lds[];
    dbg_fill_lds(p, lds);
    const TileItem t = tile_item(p, wl);
    dev_trsyrk_tile32(fronts + t.off, t.ld, t.ncs, t.nrs + 1, t.nrs, t.a, t.b, p.winv + t.woff, p.l21 + t.l21off,
                      t.nrs + 1, lds, p.junk, 0);
    kernel_release(p);
}

// ---- look-ahead schedule (CUGO_LOOKAHEAD=1; off by default, see chol_solver.h): two launches per level, the bulk of the update matrix of level
// k-1 computed WHILE level k factors its pivot blocks:
//   k_up_potrf_la : workgroups [0, npotrf)  F11 += children (extend-add part 1), L11, W   of level k
//                   the others              update tiles of level k-1 outside the lead blocks
//   k_up_lead     : workgroups [0, nlead)   per front of level k: extend-add part 3 (its lead rows),
//                                           then X and the lead block of U — all that the PARENT's
//                                           potrf waits for
//                   the others              extend-add part 4 (everything below the lead rows)
// A front's lead rows are its first la_np boundary block rows (<= 16: the ones inside the parent's
// pivot columns).  The dependent chain of a level is potrf + one lead workgroup instead of potrf +
// the whole trsm/syrk launch; sums keep their fixed order (every entry has one writer per phase).
__device__ __forceinline__ void dev_potrf_front(const CholPlanDev& p, double* __restrict__ fronts, int f,
                                                double* __restrict__ lds, int32_t* __restrict__ fail)
{
    const int ncs = 6 * p.ncb[f];
    const int ncp = pad16(ncs);
    double* Ls = lds;
    double* dinv = lds + NC_MAX * LLD;
    double* Vs = dinv + NC_MAX;
    dev_potrf_load(fronts + p.off[f], p.ldf[f], ncs, Ls, dinv);
    if (p.alias_of[f] < 0 && p.child_ptr[f + 1] > p.child_ptr[f])
    {
        dev_extend_add<true>(p, fronts, f, 0, p.ncb[f], 1, Ls, Vs + threadIdx.x);
        __syncthreads();
    }
    if (p.panel16)
    { // 16-column L D L^T panels, W built behind them (dev_potrf16)
        dev_potrf16(ncp, lds, p.winv + p.woff[f], fail);
        return;
    }
    dev_potrf_panels(ncs, Ls, dinv, fail);
    __syncthreads();
    if ((int)(threadIdx.x >> 6) < (ncp >> 4))
        inv_diag16_wave(Ls, dinv, threadIdx.x >> 6, Vs);
    __syncthreads();
    dev_winv(Ls, ncp, Vs, p.winv + p.woff[f]);
}

__global__ __launch_bounds__(BIG) void k_up_potrf_la(CholPlanDev p, double* __restrict__ fronts,
                                                     int task0, int npotrf,
                                                     const int32_t* __restrict__ wl_tiles, int tile,
                                                     int32_t* __restrict__ fail)
{
    extern __shared__ double lds[];
    if ((int)blockIdx.x < npotrf)
    {
        dev_potrf_front(p, fronts, p.task_fronts[p.task_ptr[task0 + blockIdx.x]], lds, fail);
        return;
    }
    const int32_t* it = wl_tiles + 3 * (blockIdx.x - npotrf);
    const int f = it[0];
    const int ncs = 6 * p.ncb[f], nrs = 6 * (p.nb[f] - p.ncb[f]), q = 6 * p.la_np[f];
    if (tile == 32)
        dev_trsyrk_tile32(fronts + p.off[f], p.ldf[f], ncs, nrs + 1, nrs, it[1], it[2], p.winv + p.woff[f],
                          p.l21 + p.l21off[f], nrs + 1, lds, p.junk, q);
    else
        dev_trsyrk_tile<false>(fronts + p.off[f], p.ldf[f], ncs, nrs + 1, nrs, it[1], it[2],
                               p.winv + p.woff[f], p.l21 + p.l21off[f], nrs + 1, lds, p.junk, q);
}

__global__ __launch_bounds__(BIG) void k_up_lead(CholPlanDev p, double* __restrict__ fronts,
                                                 const int32_t* __restrict__ wl_lead, int nlead,
                                                 const int32_t* __restrict__ wl_eap, int neap,
                                                 const int32_t* __restrict__ wl_eab)
{
    extern __shared__ double lds[];
    if ((int)blockIdx.x >= nlead)
    {
        const int b = blockIdx.x - nlead;
        const int32_t* it = b < neap ? wl_eap + 3 * b : wl_eab + 3 * (b - neap);
        dev_extend_add(p, fronts, it[0], it[1], it[2], 4);
        return;
    }
    const int f = wl_lead[3 * blockIdx.x];
    const int ncb = p.ncb[f], np = p.la_np[f];
    const int ncs = 6 * ncb, nrs = 6 * (p.nb[f] - ncb), q = 6 * np;
    if (p.alias_of[f] < 0 && p.child_ptr[f + 1] > p.child_ptr[f])
    {
        dev_extend_add(p, fronts, f, 0, ncb + np, 3);
        __threadfence_block();
        __syncthreads();
    }
    dev_trsyrk_tile<true>(fronts + p.off[f], p.ldf[f], ncs, q, q, q > 64 ? 1 : 0, 0, p.winv + p.woff[f],
                          p.l21 + p.l21off[f], nrs + 1, lds, p.junk, 0);
}

// ---- two-phase form of the tile work, for levels with more 64x64 tiles than the chip has CUs: there
// the fused tile kernel is bound by throughput, and two thirds of its matrix-core work is the X = B W^T
// of its two row tiles, recomputed by every tile of the row / column.  k_up_trsm solves every 64-row
// tile ONCE (X in place of B in the front, and into the compact L21 buffer), k_up_syrk then forms
// U(ti,tj) -= X_i X_j^T with K staged in two halves, so that two workgroups fit a CU.
constexpr int KC_SYRK2 = 48;
__global__ __launch_bounds__(BIG) void k_up_trsm(CholPlanDev p, double* __restrict__ fronts,
                                                 const int32_t* __restrict__ wl)
{
#ifdef CUGO_DEBUG_HOOKS
    if (p.dbg_skip_wg == (int)blockIdx.x)
        return; // (fault injection, see DbgSkip)
#endif
    kernel_acquire(p);
    extern __shared__ double lds[];
    dbg_fill_lds(p, lds);
    const TileItem t = tile_item(p, wl);
    dev_trsm_w(fronts + t.off, t.ld, t.ncs, (long)t.ncs + t.a, t.b, p.winv + t.woff, lds, p.junk, p.l21 + t.l21off,
               t.nrs + 1);
    kernel_release(p);
}

__global__ __launch_bounds__(BIG) void k_up_syrk(CholPlanDev p, double* __restrict__ fronts,
                                                 const int32_t* __restrict__ wl)
{
#ifdef CUGO_DEBUG_HOOKS
    if (p.dbg_skip_wg == (int)blockIdx.x)
        return; // (fault injection, see DbgSkip)
#endif
    kernel_acquire(p);
    extern __shared__ double lds[];
    dbg_fill_lds(p, lds);
    const TileItem t = tile_item(p, wl);
    dev_syrk_tiles<KC_SYRK2>(fronts + t.off, t.ld, t.ncs, t.nrs + 1, t.nrs, t.a, t.a + 1, t.b, lds, p.junk);
    kernel_release(p);
}

// the ancestor part of a front's backward mat-vec, one launch ahead of the front itself:
// v_j = y_j - sum_{i >= 6*npb} L21[i,j] x_R[i] for the 16 columns j0.. (one wave per column, its
// 64 lanes stride the rows), parked in xnew at the front's own positions
__device__ __forceinline__ void dev_backward_ahead(const CholPlanDev& p, const int32_t* __restrict__ ft, double* __restrict__ lds,
                                                   double* __restrict__ xnew)
{
    // the item's 64-byte record (CholPlanDev::fat): {front, first column, -, 6 ncb, 6 (nb - ncb), bw_np, col0,
    // rows_ptr, off, ldf, woff, l21off} — one load instead of item -> front -> six per-front arrays
    const int j0 = ft[1];
    const int ncs = ft[3], nrs = ft[4];
    const int r0 = 6 * ft[5]; // first row of the ancestor part
    const double* L = p.l21 + reinterpret_cast<const long*>(ft + 8)[3];
    const long ldl = nrs + 1;
    const int c0 = ft[6];
    const int32_t* rows = p.rows + ft[7];
    double* xr = lds;
    for (int i = r0 + threadIdx.x; i < nrs; i += blockDim.x)
    {
        const int ib = i / 6;
        xr[i] = xnew[6L * rows[ib] + (i - 6 * ib)];
    }
    TILE_SYNC();
    const int j = j0 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j < ncs)
    {
        const double* col = L + (long)j * ldl;
        const double y = col[nrs];
        double s = 0;
        for (int i = r0 + lane; i < nrs; i += 512)
        { // eight independent loads in flight per lane
            double a[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                a[u] = col[min(i + 64 * u, nrs - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++)
                s += (i + 64 * u < nrs) ? a[u] * xr[i + 64 * u] : 0.0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            s += __shfl_xor(s, off, 64);
        if (lane == 0)
            xnew[6L * c0 + j] = y - s;
    }
}

__global__ __launch_bounds__(BIG) void k_backward_stage(CholPlanDev p,
                                                        const double* __restrict__ fronts,
                                                        int task0, int ntasks,
                                                        const int32_t* __restrict__ wl_gemv,
                                                        double* __restrict__ xnew,
                                                        double* __restrict__ xout)
{
#ifdef CUGO_DEBUG_HOOKS
    if (p.dbg_skip_wg == (int)blockIdx.x)
        return; // (fault injection, see DbgSkip)
#endif
    kernel_acquire(p);
    extern __shared__ double lds[];
    dbg_fill_lds(p, lds);
    if ((int)blockIdx.x >= ntasks)
    { // ahead-of-time mat-vec of a child of this level's fronts
        dev_backward_ahead(p, p.fat + 16 * ((int)((wl_gemv - p.wl_base) / 3) + ((int)blockIdx.x - ntasks)), lds, xnew);
        kernel_release(p);
        return;
    }
    stamp(3, 0);
    const int task = task0 + blockIdx.x;
    const int32_t* tm = p.tmeta + cugo_k::TMETA * task;
    if (tm[0] == 1)
        dev_backward(p, fronts, front_view(tm), lds, xnew, xout, tm + 20);
    else
        for (int fi = p.task_ptr[task + 1] - 1; fi >= p.task_ptr[task]; fi--)
            dev_backward(p, fronts, front_view(p, p.task_fronts[fi]), lds, xnew, xout);
    stamp(3, 7);
    kernel_release(p);
}

// x[perm[j]] = xnew[j] for all block rows (after the solution ranges of other ranks' subtrees have arrived)
__global__ __launch_bounds__(CBS) void k_unpermute(CholPlanDev p, const double* __restrict__ xnew, double* __restrict__ xout)
{
    const int i = blockIdx.x * CBS + threadIdx.x;
    if (i < 6 * p.n)
        xout[6L * p.perm[i / 6] + i % 6] = xnew[i];
}

// Ownership-keyed exchange of the Schur system (chol_symbolic.h: CholPlan::xs_off): sys = [Hsc (36 B) | bsc (6 n)] in
// block / pose order  <->  the packed buffer whose segment r holds what rank r's fronts assemble and whose tail holds
// what the replicated top assembles.  Pack copies every unit (this rank's partial sums of all of them go out); unpack
// brings back only the units this rank will assemble: offsets in [lo, hi) — its own segment — or from top0 on.
__global__ __launch_bounds__(CBS) void k_xs_pack(const int64_t* __restrict__ off, int B, int n,
                                                 const double* __restrict__ sys, double* __restrict__ xbuf)
{
    const long i = (long)blockIdx.x * CBS + threadIdx.x;
    const long nH = 36L * B;
    if (i >= nH + 6L * n)
        return;
    const long dst = i < nH ? off[i / 36] + i % 36 : off[B + (i - nH) / 6] + (i - nH) % 6;
    xbuf[dst] = sys[i];
}
__global__ __launch_bounds__(CBS) void k_xs_unpack(const int64_t* __restrict__ off, int B, int n,
                                                   const double* __restrict__ xbuf, double* __restrict__ sys,
                                                   long lo, long hi, long top0)
{
    const long i = (long)blockIdx.x * CBS + threadIdx.x;
    const long nH = 36L * B;
    if (i >= nH + 6L * n)
        return;
    const long o = i < nH ? off[i / 36] : off[B + (i - nH) / 6];
    if ((o >= lo && o < hi) || o >= top0)
        sys[i] = xbuf[o + (i < nH ? i % 36 : (i - nH) % 6)];
}

// the zero-pivot flag (int32 in an 8-byte slot) as a double in the same slot: ranks that factor different
// subtrees see different flags, and the flag then rides in the sum all-reduce of the trial's scalars
__global__ void k_flag_to_double(int32_t* __restrict__ flag)
{
    const int v = *flag;
    *reinterpret_cast<double*>(flag) = v ? 1.0 : 0.0;
}

#ifdef CUGO_DEBUG_HOOKS
// Fault injection (diagnosis; CUGO_DEBUG_SKIP=call:launch:workgroup, read by the solver): ONE
// workgroup of ONE launch of one factorisation returns at once, so everything it would have written keeps the value
// the previous factorisation left there — the supposed failure, made on purpose, to compare its results with the
// alternates the rare deviation produces.  The launches of a factorisation are counted in queueing order.
struct DbgSkip
{
    int launch = 0, target_launch = -1, target_wg = -1;
    std::FILE* dump = nullptr;
};
static DbgSkip g_skip;
static inline CholPlanDev with_lds(const CholPlanDev& p, size_t lds_bytes, const char* name, int grid, int first = 0)
{ // (the plan is a kernel argument passed by value: the copy carries this launch's LDS size for dbg_fill_lds)
    CholPlanDev q = p;
    q.lds_doubles = (int)(lds_bytes / sizeof(double));
    q.dbg_skip_wg = -1;
    static int tile_delay_on_device = 0;
    if (p.dbg_delay != tile_delay_on_device)
    { // (a blocking copy: diagnosis only)
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_tile_delay), &p.dbg_delay, sizeof(int));
        tile_delay_on_device = p.dbg_delay;
    }
    if (g_skip.dump || g_skip.target_launch >= 0)
    {
        if (g_skip.dump)
            std::fprintf(g_skip.dump, "%d %s %d %d\n", g_skip.launch, name, grid, first);
        if (g_skip.launch == g_skip.target_launch)
            q.dbg_skip_wg = g_skip.target_wg;
        g_skip.launch++;
    }
    return q;
}
#else
// (product build: the plan goes to the kernels as it is — no per-launch copy, no hooks)
static inline const CholPlanDev& with_lds(const CholPlanDev& p, size_t, const char*, int, int = 0) { return p; }
#endif
void ensure_lds(const void* fn, size_t bytes)
{
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

} // namespace

namespace cugo_k
{
#ifdef CUGO_DEBUG_HOOKS
void chol_dbg_skip_begin(int target_launch, int target_wg, const char* dump_path)
{
    if (g_skip.dump)
        std::fclose(g_skip.dump), g_skip.dump = nullptr;
    g_skip.launch = 0, g_skip.target_launch = target_launch, g_skip.target_wg = target_wg;
    if (dump_path)
        g_skip.dump = std::fopen(dump_path, "w");
}
void chol_dbg_skip_end()
{
    if (g_skip.dump)
        std::fclose(g_skip.dump), g_skip.dump = nullptr;
    g_skip.target_launch = -1;
}
#endif

size_t chol_lds_factor_bytes(int nc_max)
{ // subtree / potrf kernels: L11 + dinv + inverted diagonal blocks + one trsm B tile, or the syrk panels
    (void)nc_max; // fixed layout (LLD): see dev_load_l11
    const size_t trsm = (size_t)NC_MAX * LLD + NC_MAX + std::max((size_t)(NC_MAX >> 4) * (16 * 17), (size_t)NC_MAX * PSTB);
    const size_t syrk = (size_t)syrk_lds();
    return (std::max(trsm, syrk) + 8) * sizeof(double);
}
size_t chol_lds_potrf_bytes()
{
    return std::max((size_t)NC_MAX * LLD + NC_MAX + (NC_MAX >> 4) * (16 * 17) + 8, (size_t)p16::LDS_DOUBLES + 8) *
           sizeof(double);
}
int chol_max_pivot_cols() { return NC_MAX; }
size_t chol_lds_backward_bytes(int nc_max, long ld_max)
{
    const size_t ncp = (size_t)((nc_max + 15) & ~15);
    return (ncp + (size_t)ld_max + 8) * sizeof(double);
}

void set_debug_stamps(long long* d_buf)
{
#ifdef CUGO_STAMPS
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_buf, sizeof(d_buf));
#else
    (void)d_buf;
#endif
}

void launch_chol_assemble(hipStream_t s, const CholPlanDev& p, double* d_fronts,
                          size_t front_doubles, const double* d_Hsc, double lambda,
                          const double* d_bsc, int32_t* d_fail, const int32_t* d_clear_items, int nclear,
                          const int32_t* d_asm_items, int nasm)
{
    if (d_asm_items && nasm > 0)
    {
        const long n = 36L * p.n_hsc_blocks;
        const int nblk = (int)((n + CBS - 1) / CBS), nrhs = std::max(1, (6 * p.n + CBS - 1) / CBS);
        CUGO_LAUNCH(k_assemble_fronts, dim3(nasm + nblk + nrhs), dim3(CBS), 0, s, p, d_fronts, d_Hsc, lambda, d_bsc,
                    d_fail, (int)((d_asm_items - p.wl_base) / 3), nasm, nblk);
        return;
    }
    if (nclear > 0)
        CUGO_LAUNCH(k_clear_fronts, dim3(nclear), dim3(CBS), 0, s, p, d_fronts, d_clear_items);
    else
        (void)hipMemsetAsync(d_fronts, 0, front_doubles * sizeof(double), s);
    const long n = 36L * p.n_hsc_blocks;
    const int nblk = (int)((n + CBS - 1) / CBS), nrhs = std::max(1, (6 * p.n + CBS - 1) / CBS);
    CUGO_LAUNCH(k_assemble_blocks, dim3(nblk + nrhs), dim3(CBS), 0, s, p, d_fronts, d_Hsc, lambda, d_bsc, d_fail,
                nblk);
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

void launch_chol_two_phase(hipStream_t s, const CholPlanDev& p, double* d_fronts, const int32_t* d_trsm, int ntrsm,
                           const int32_t* d_syrk, int nsyrk)
{
    const size_t lds_t = (size_t)NC_MAX * PSTB * sizeof(double), lds_s = (size_t)2 * KC_SYRK2 * PST * sizeof(double);
    if (ntrsm > 0)
    {
        ensure_lds(reinterpret_cast<const void*>(k_up_trsm), lds_t);
        CUGO_LAUNCH(k_up_trsm, dim3(ntrsm), dim3(BIG), lds_t, s, with_lds(p, lds_t, "k_up_trsm", ntrsm), d_fronts, d_trsm);
    }
    if (nsyrk > 0)
    {
        ensure_lds(reinterpret_cast<const void*>(k_up_syrk), lds_s);
        CUGO_LAUNCH(k_up_syrk, dim3(nsyrk), dim3(BIG), lds_s, s, with_lds(p, lds_s, "k_up_syrk", nsyrk), d_fronts, d_syrk);
    }
}

void launch_chol_upper_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                             int ntasks, const int32_t* d_wl, int eap0, int neap, int ea0, int nea,
                             int sy0, int nsy, int tile, size_t lds_bytes, int32_t* d_fail, double* dbg_line,
                             double* dbg_scratch)
{
    (void)lds_bytes;
    if (ntasks <= 0)
        return;
    ensure_lds(reinterpret_cast<const void*>(k_up_potrf), chol_lds_potrf_bytes());
    CUGO_LAUNCH(k_up_potrf, dim3(ntasks + neap + nea), dim3(BIG), chol_lds_potrf_bytes(), s, with_lds(p, chol_lds_potrf_bytes(), "k_up_potrf", ntasks + neap + nea, ntasks), d_fronts,
                task0, ntasks, d_wl + 3L * eap0, neap, d_wl + 3L * ea0, d_fail);
#ifdef CUGO_DEBUG_HOOKS
    if (dbg_line) // (CUGO_DEBUG_STALE: the tile launch sees the line as it was before this potrf launch ...
        launch_swap16(s, dbg_line, dbg_scratch);
#else
    (void)dbg_line, (void)dbg_scratch;
#endif
    if (tile == 0)
        return; // two-phase level: the caller queues launch_chol_two_phase
    if (nsy > 0 && tile == 32)
    {
        const size_t lds32 = (2 * KC_SYRK * TPST32 + 21 * 256) * sizeof(double);
        ensure_lds(reinterpret_cast<const void*>(k_up_trsyrk32), lds32);
        CUGO_LAUNCH(k_up_trsyrk32, dim3(nsy), dim3(BIG), lds32, s, with_lds(p, lds32, "k_up_trsyrk32", nsy), d_fronts, d_wl + 3L * sy0);
    }
    else if (nsy > 0)
    {
        ensure_lds(reinterpret_cast<const void*>(k_up_trsyrk), trsyrk_lds() * sizeof(double));
        CUGO_LAUNCH(k_up_trsyrk, dim3(nsy), dim3(BIG), trsyrk_lds() * sizeof(double), s,
                    with_lds(p, trsyrk_lds() * sizeof(double), "k_up_trsyrk", nsy), d_fronts, d_wl + 3L * sy0);
    }
#ifdef CUGO_DEBUG_HOOKS
    if (dbg_line) // ... and everything later sees what the potrf launch wrote)
        launch_swap16(s, dbg_line, dbg_scratch);
#endif
}

void launch_chol_potrf_la(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0, int ntasks,
                          const int32_t* d_tiles, int ntiles, int tile, int32_t* d_fail)
{
    if (ntasks + ntiles <= 0)
        return;
    const size_t lds32 = (2 * KC_SYRK * TPST32 + 21 * 256) * sizeof(double);
    const size_t bytes = std::max(chol_lds_potrf_bytes(),
                                  ntiles > 0 ? (tile == 32 ? lds32 : trsyrk_lds() * sizeof(double)) : size_t(0));
    ensure_lds(reinterpret_cast<const void*>(k_up_potrf_la), std::max(chol_lds_potrf_bytes(), trsyrk_lds() * sizeof(double)));
    CUGO_LAUNCH(k_up_potrf_la, dim3(ntasks + ntiles), dim3(BIG), bytes, s, p, d_fronts, task0, ntasks,
                d_tiles, tile, d_fail);
}

void launch_chol_lead(hipStream_t s, const CholPlanDev& p, double* d_fronts, const int32_t* d_lead,
                      int nlead, const int32_t* d_eap, int neap, const int32_t* d_eab, int neab)
{
    if (nlead + neap + neab <= 0)
        return;
    ensure_lds(reinterpret_cast<const void*>(k_up_lead), trsyrk_lds() * sizeof(double));
    CUGO_LAUNCH(k_up_lead, dim3(nlead + neap + neab), dim3(BIG), trsyrk_lds() * sizeof(double), s, p,
                d_fronts, d_lead, nlead, d_eap, neap, d_eab);
}

#ifdef CUGO_DEBUG_HOOKS // diagnosis kernels: only in libcugo_hip_hooks.so
__global__ __launch_bounds__(256) void k_hash_words(const unsigned long long* __restrict__ p, size_t n,
                                                    unsigned long long* __restrict__ out)
{
    unsigned long long h = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        h += p[i] * (2 * i + 1); // (position-weighted: a swap of two words changes the sum)
    atomicAdd(out, h);
}
__global__ void k_swap16(double* __restrict__ a, double* __restrict__ b)
{
    const double x = a[threadIdx.x], y = b[threadIdx.x];
    a[threadIdx.x] = y, b[threadIdx.x] = x;
}
void launch_swap16(hipStream_t s, double* line, double* scratch) { hipLaunchKernelGGL(k_swap16, dim3(1), dim3(16), 0, s, line, scratch); }
__global__ void k_nop() {}
void launch_nop(hipStream_t s) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s); }
void launch_hash_words(hipStream_t s, const void* p, size_t n_words, unsigned long long* out)
{
    if (n_words == 0)
        return;
    const int nb = (int)std::min<size_t>(512, (n_words + 255) / 256);
    hipLaunchKernelGGL(k_hash_words, dim3(nb), dim3(256), 0, s, static_cast<const unsigned long long*>(p), n_words, out);
}
#endif
void launch_flag_to_double(hipStream_t s, int32_t* d_flag) { CUGO_LAUNCH(k_flag_to_double, dim3(1), dim3(1), 0, s, d_flag); }

void launch_xs_pack(hipStream_t s, const int64_t* d_off, int B, int n, const double* d_sys, double* d_xbuf)
{
    const long total = 36L * B + 6L * n;
    if (total > 0)
        CUGO_LAUNCH(k_xs_pack, dim3((unsigned)((total + CBS - 1) / CBS)), dim3(CBS), 0, s, d_off, B, n, d_sys, d_xbuf);
}
void launch_xs_unpack(hipStream_t s, const int64_t* d_off, int B, int n, const double* d_xbuf, double* d_sys, long lo,
                      long hi, long top0)
{
    const long total = 36L * B + 6L * n;
    if (total > 0)
        CUGO_LAUNCH(k_xs_unpack, dim3((unsigned)((total + CBS - 1) / CBS)), dim3(CBS), 0, s, d_off, B, n, d_xbuf, d_sys,
                    lo, hi, top0);
}

void launch_chol_unpermute(hipStream_t s, const CholPlanDev& p, const double* d_xnew, double* d_x)
{
    if (p.n > 0)
        CUGO_LAUNCH(k_unpermute, dim3((6 * p.n + CBS - 1) / CBS), dim3(CBS), 0, s, p, d_xnew, d_x);
}

void launch_chol_backward_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                                int ntasks, size_t lds_bytes, double* d_xnew, double* d_x,
                                const int32_t* d_wl_gemv, int ngemv)
{
    if (ntasks <= 0)
        return;
    ensure_lds(reinterpret_cast<const void*>(k_backward_stage), lds_bytes);
    CUGO_LAUNCH(k_backward_stage, dim3(ntasks + ngemv), dim3(BIG), lds_bytes, s, with_lds(p, lds_bytes, "k_backward_stage", ntasks + ngemv, ntasks), d_fronts, task0,
                       ntasks, d_wl_gemv, d_xnew, d_x);
}

} // namespace cugo_k
