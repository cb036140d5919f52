// Edge / landmark / pose kernels of the BA hot path for gfx950 (wave64, fp64).
//
// Design:
//  * edges are landmark-major, mono+stereo merged, streamed with lane <-> edge so every
//    per-edge array is read/written fully coalesced;
//  * no floating-point atomics anywhere: Hll/bl are summed per landmark, Hpp/bp and the
//    diagonal Schur blocks per pose by one workgroup each, off-diagonal Schur blocks by one
//    wave each over a precomputed contribution list — every sum has a fixed order, so runs
//    are bit-reproducible (the reference accumulates with fp64 atomicAdd, F8 in SURVEY.md);
//  * damping is applied on the fly, Hpp/Hll stay undamped (no addLambda/restoreDiagonal).
//
// ref: src/cuda/cuda_block_solver.cu — kernels at 1060 (errors), 1152 (quadratic form),
// 1223 (max diagonal), 1286-1345 (Schur), 1419-1490 (back-substitution, update, scale).
#include "ba_math.h"
#include "kernels.h"

using namespace cugo_dev;

namespace
{

constexpr int BS = 256;

struct EV
{
    int E, Pall, Lall, P, L;
    const int32_t* pose;
    const int32_t* lm;
    const double* meas;
    const double* omega;
    int n_omega;
    const uint8_t* flags;
    const uint16_t* cam;
    const double* cams;
    int n_cams;
    const int32_t* lm_ptr;
    const int32_t* pose_ptr;
    const int32_t* pose_edge;
};

EV make_ev(const cugo_edges& e)
{
    EV v;
    v.E = e.n_edges, v.Pall = e.n_poses_total, v.Lall = e.n_landmarks_total;
    v.P = e.n_poses_free, v.L = e.n_landmarks_free;
    v.pose = e.d_pose, v.lm = e.d_lm, v.meas = e.d_meas, v.omega = e.d_omega;
    v.n_omega = e.n_omega, v.flags = e.d_flags, v.cam = e.d_cam, v.cams = e.d_cams;
    v.n_cams = e.n_cams, v.lm_ptr = e.d_lm_ptr, v.pose_ptr = e.d_pose_ptr;
    v.pose_edge = e.d_pose_edge;
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}

// deterministic block sum (BS threads); result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sm)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0)
        sm[wv] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0)
    {
        for (int i = 0; i < (int)(blockDim.x >> 6); i++)
            r += sm[i];
    }
    __syncthreads();
    return r;
}

// Workgroups b and b+8 land on the same XCD (observed round-robin dispatch; speed only, never
// correctness) and share its 4 MiB L2.  k_hsc_diag gives every XCD a CONTIGUOUS range of poses:
// neighbouring poses read the same landmarks' 144-B blocks (adjacent slots, shared cache lines),
// so a line fetched once serves the neighbours out of L2 (57 -> 47 us on the kitti_00 shape).
// Launch 8*ceil(n/8) workgroups; returns the work item of this workgroup (>= n: none).
__device__ __forceinline__ int xcd_contiguous_item(int n)
{
    const int per = (n + 7) >> 3;
    return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}
static inline int xcd_grid(int n) { return 8 * ((n + 7) / 8); }

// Element type S of the two per-edge block streams, Hpl [E][18] and T = Hpl invHll [E][18]:
// double, or float in the fp32-internal mode (BASELINE config 5; cugo_edges.block_f32).  Those
// two arrays are what the Schur kernels move, so float storage halves their traffic; every
// value is widened on load and all arithmetic and every accumulator stays fp64.  The blocks are
// moved as pairs of consecutive elements: 16-B accesses for double, 8-B for float, the same
// lane <-> pair assignment in both.
template <typename S>
struct BlockPair;
template <>
struct BlockPair<double>
{
    using type = double2;
};
template <>
struct BlockPair<float>
{
    using type = float2;
};
template <typename S>
__device__ __forceinline__ double2 ld_pair(const S* __restrict__ base, size_t pair)
{
    const typename BlockPair<S>::type v = reinterpret_cast<const typename BlockPair<S>::type*>(base)[pair];
    return make_double2((double)v.x, (double)v.y);
}
template <typename S>
__device__ __forceinline__ void st_pair(S* __restrict__ base, size_t pair, double a, double b)
{
    typename BlockPair<S>::type v;
    v.x = (S)a, v.y = (S)b;
    reinterpret_cast<typename BlockPair<S>::type*>(base)[pair] = v;
}

// a wave's own LDS writes become visible to its other lanes (no workgroup barrier needed)
__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

struct Robust2
{
    Robust m, s; // mono / stereo edge sets
};

struct EdgeIn
{
    int ip, il;
    double mu, mv, mr, omega;
    bool stereo;
    const double* cam;
};

__device__ __forceinline__ EdgeIn load_edge(const EV& ev, int e, uint8_t fl)
{
    EdgeIn in;
    in.ip = ev.pose[e];
    in.il = ev.lm[e];
    in.stereo = (fl & CUGO_EDGE_STEREO) != 0;
    in.mu = ev.meas[e];
    in.mv = ev.meas[(size_t)ev.E + e];
    in.mr = in.stereo ? ev.meas[2 * (size_t)ev.E + e] : 0.0;
    in.omega = ev.n_omega > 1 ? ev.omega[e] : ev.omega[0];
    in.cam = ev.n_cams > 1 ? ev.cams + 5 * (int)ev.cam[e] : ev.cams;
    return in;
}

// ---------------------------------------------------------------- reductions -----------
// one workgroup, fixed order: four partial sums per thread (four loads in flight), then the
// block tree.  1024 threads: the 5M-edge graph has 19.5k partials per pass
constexpr int SP_BS = 1024;
__global__ __launch_bounds__(SP_BS) void k_sum_partials(const double* __restrict__ part, int n,
                                                        double* __restrict__ out)
{
    __shared__ double sm[SP_BS / 64];
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    int i = threadIdx.x;
    for (; i + 3 * SP_BS < n; i += 4 * SP_BS)
    {
        const double a = part[i], b = part[i + SP_BS], c = part[i + 2 * SP_BS], d = part[i + 3 * SP_BS];
        v0 += a, v1 += b, v2 += c, v3 += d;
    }
    for (; i < n; i += SP_BS)
        v0 += part[i];
    const double v = block_sum((v0 + v1) + (v2 + v3), sm);
    if (threadIdx.x == 0)
        out[0] = v;
}

// the two reductions at the end of an LM trial in ONE launch: workgroup 0 sums the chi2 partials of
// the error pass, workgroup 1 the scale partials of the update pass — each exactly as k_sum_partials
// does (same order, same result) — and both also deposit their value, workgroup 0 the factorisation
// flag too, in a pinned host block: no device-to-host copy is queued behind them
__global__ __launch_bounds__(SP_BS) void k_sum_partials2(const double* __restrict__ partA, int nA,
                                                         const double* __restrict__ partB, int nB,
                                                         double* __restrict__ out, const double* __restrict__ flag,
                                                         double* __restrict__ host_out, double seq,
                                                         unsigned* __restrict__ done)
{
    __shared__ double sm[SP_BS / 64];
    const double* part = blockIdx.x == 0 ? partA : partB;
    const int n = blockIdx.x == 0 ? nA : nB;
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    int i = threadIdx.x;
    for (; i + 3 * SP_BS < n; i += 4 * SP_BS)
    {
        const double a = part[i], b = part[i + SP_BS], c = part[i + 2 * SP_BS], d = part[i + 3 * SP_BS];
        v0 += a, v1 += b, v2 += c, v3 += d;
    }
    for (; i < n; i += SP_BS)
        v0 += part[i];
    const double v = block_sum((v0 + v1) + (v2 + v3), sm);
    if (threadIdx.x == 0)
    {
        out[blockIdx.x] = v;
        if (host_out)
        {
            host_out[blockIdx.x] = v;
            if (blockIdx.x == 0)
                host_out[2] = *flag; // 8 bytes holding the int32 zero-pivot flag
            // the trial's sequence number follows the three words (system-scope release; the second of the two
            // workgroups to get here writes it): the host checks it after its wait, so a wait that returned
            // early could never hand it the previous trial's numbers
            __threadfence_system();
            if (atomicAdd(done, 1u) == 1u)
            {
                *done = 0u;
                __hip_atomic_store(host_out + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ __launch_bounds__(BS) void k_max_partials(const double* __restrict__ part, int n,
                                                     double* __restrict__ out)
{
    __shared__ double sm[BS];
    double v = 0;
    for (int i = threadIdx.x; i < n; i += BS)
        v = fmax(v, part[i]);
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int s = BS / 2; s > 0; s >>= 1)
    {
        if ((int)threadIdx.x < s)
            sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[0] = sm[0];
}

// ---------------------------------------------------------------- errors ---------------
// one lane per edge; chi2 partial per workgroup (ref: computeActiveErrorsKernel .cu:1060 +
// computeChiValueKernel .cu:1112, fused; errors/Xc are not materialised)
__global__ __launch_bounds__(BS) void k_errors(EV ev, const double* __restrict__ poses,
                                               const double* __restrict__ lms, Robust2 rk,
                                               double* __restrict__ partials)
{
    __shared__ double sm[BS / 64];
    const int e = blockIdx.x * BS + threadIdx.x;
    double chi = 0;
    if (e < ev.E)
    {
        const uint8_t fl = ev.flags[e];
        if (!(fl & CUGO_EDGE_INACTIVE))
        {
            const EdgeIn in = load_edge(ev, e, fl);
            EdgeGeom g;
            edge_residual(poses + 7 * (size_t)in.ip, lms + 3 * (size_t)in.il, in.mu, in.mv, in.mr,
                          in.stereo, in.omega, in.cam, in.stereo ? rk.s : rk.m, g);
            chi = g.chi;
        }
    }
    chi = block_sum(chi, sm);
    if (threadIdx.x == 0)
        partials[blockIdx.x] = chi;
}

// chi_e = rho(omega |e|^2) per edge slot (0 for inactive slots): only used by the outlier
// rejection at the end of optimize() (ref: d_chiValues read by computeOutliersKernel .cu:1135)
__global__ __launch_bounds__(BS) void k_edge_chi(EV ev, const double* __restrict__ poses,
                                                 const double* __restrict__ lms, Robust2 rk,
                                                 double* __restrict__ chi_out)
{
    const int e = blockIdx.x * BS + threadIdx.x;
    if (e >= ev.E)
        return;
    const uint8_t fl = ev.flags[e];
    double chi = 0;
    if (!(fl & CUGO_EDGE_INACTIVE))
    {
        const EdgeIn in = load_edge(ev, e, fl);
        EdgeGeom g;
        edge_residual(poses + 7 * (size_t)in.ip, lms + 3 * (size_t)in.il, in.mu, in.mv, in.mr,
                      in.stereo, in.omega, in.cam, in.stereo ? rk.s : rk.m, g);
        chi = g.chi;
    }
    chi_out[e] = chi;
}

// ---------------------------------------------------------------- build: edges + landmarks
// One lane per edge: residual and Jacobians are evaluated ONCE and feed both
//   Hpl[e] = w JP^T JL (6x3 col-major, global) and the chi2 partials, and
//   the edge's landmark contribution w JL^T [JL | e] (9 values, LDS).
// Edges are landmark-major, so the edges of a landmark are consecutive lanes: the lane that
// holds a landmark's FIRST edge sums the contributions of its edges from LDS in edge order
// (fixed order, no atomics) and writes Hll[l], bl[l].  The engine pads the edge array with
// inactive edges so that no landmark straddles a 256-edge block; for arbitrary layouts
// (kernel-level C ABI) the owner recomputes the few edges that lie beyond its block.
// Thread i of the grid also zero-fills landmark i if it has no edge at all.
struct LmContrib
{
    double h00, h01, h02, h11, h12, h22, b0, b1, b2;
};
__device__ __forceinline__ LmContrib lm_contrib(const double JL[3][3], const EdgeGeom& g, bool stereo)
{
    const int dim = stereo ? 3 : 2;
    double s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0, t0 = 0, t1 = 0, t2 = 0;
    for (int m = 0; m < dim; m++)
    {
        s00 += JL[m][0] * JL[m][0];
        s01 += JL[m][0] * JL[m][1];
        s02 += JL[m][0] * JL[m][2];
        s11 += JL[m][1] * JL[m][1];
        s12 += JL[m][1] * JL[m][2];
        s22 += JL[m][2] * JL[m][2];
        t0 += JL[m][0] * g.e[m];
        t1 += JL[m][1] * g.e[m];
        t2 += JL[m][2] * g.e[m];
    }
    return LmContrib{g.w * s00, g.w * s01, g.w * s02, g.w * s11, g.w * s12, g.w * s22,
                     g.w * t0, g.w * t1, g.w * t2};
}

// (144 VGPRs: three waves per SIMD.  Capped at 128 — four waves — the kernel spills ~230 bytes per lane and takes
// 75 instead of 56 us: 11.43 vs 11.17 ms per step, 38.9 vs 38.2 ms on the 10k-pose graph.)
template <typename S>
__global__ __launch_bounds__(BS) void k_build_edges(EV ev, const double* __restrict__ poses,
                                                    const double* __restrict__ lms, Robust2 rk,
                                                    S* __restrict__ Hpl,
                                                    double* __restrict__ Hll,
                                                    double* __restrict__ bl,
                                                    double* __restrict__ rec,
                                                    double* __restrict__ partials,
                                                    double fuse_lambda, double* __restrict__ invHll,
                                                    S* __restrict__ T, double* __restrict__ lmrec)
{
    // fuse_lambda >= 0 (the engine, whose slot layout keeps a landmark inside one workgroup, from the
    // second LM iteration on: the damping of the first trial is known when the build is queued): this
    // pass also leaves invHll = (Hll + lambda I)^-1 and T = Hpl invHll — exactly what k_schur_edges
    // would compute from the arrays written here, without reading the Hpl stream again.
    // lmrec != nullptr (with fuse and T): ONE block stream instead of two.  With Hll + lambda I = L L^T the pass
    // writes G = Hpl L^-T into T and neither Hpl nor invHll: Hsc_ij = - sum G_i G_j^T (both operands of the
    // off-diagonal kernel out of one array), dx_l = L^-T (y - sum G_e^T dx_p) with y = L^-1 bl (k_backsubst_landmarks),
    // and the pose pass (k_pose_schur) needs L^-1 and y only: the landmark's line lmrec[16 l] = {L^-1 (6), y (3)}.
    const bool fuse = fuse_lambda >= 0.0;
    const bool gform = fuse && T != nullptr && lmrec != nullptr;
    __shared__ double sm[BS / 64];
    // 36 KB used twice: first the landmark contributions cs[9][BS] and the per-edge records
    // rs_[BS*9] (9-double lane stride: conflict-free both ways), at the end the block's 256 Hpl
    // blocks on their way to coalesced stores
    __shared__ double2 pool2[BS * 9 + 1];
    double(*cs)[BS] = reinterpret_cast<double(*)[BS]>(pool2);
    double* rs_ = reinterpret_cast<double*>(pool2) + 9 * BS;
    // per landmark owner slot: the six entries of invHll (7: bank spread).  Lives in the record area once the
    // records have left for global memory — 36 KB of LDS per workgroup instead of 50: four workgroups per CU
    double(*ivs)[7] = reinterpret_cast<double(*)[7]>(rs_);
    const int e = blockIdx.x * BS + threadIdx.x;
    double chi = 0;
    double H[18];
#pragma unroll
    for (int i = 0; i < 18; i++)
        H[i] = 0;
    LmContrib lc = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int l = -1;
    // record for the pose pass (k_build_poses): Xc, e, w and {camera index, stereo bit}; w = 0 for
    // an inactive slot.  The pose pass then needs ONE 64-byte line per edge instead of ~8 gathers.
    double r8[8] = {0, 0, 1, 0, 0, 0, 0, 0};
    if (e < ev.E)
    {
        const uint8_t fl = ev.flags[e];
        l = ev.lm[e];
        if (!(fl & CUGO_EDGE_INACTIVE))
        {
            const EdgeIn in = load_edge(ev, e, fl);
            const double* pose = poses + 7 * (size_t)in.ip;
            EdgeGeom g;
            edge_residual(pose, lms + 3 * (size_t)in.il, in.mu, in.mv, in.mr, in.stereo, in.omega,
                          in.cam, in.stereo ? rk.s : rk.m, g);
            chi = g.chi;
            r8[0] = g.Xc[0], r8[1] = g.Xc[1], r8[2] = g.Xc[2];
            r8[3] = g.e[0], r8[4] = g.e[1], r8[5] = g.e[2], r8[6] = g.w;
            // {camera index, stereo bit, "takes part in the Schur complement" bit, landmark} (k_pose_schur)
            const bool schur_act = !(fl & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P)) && l < ev.L;
            r8[7] = __longlong_as_double((long long)(ev.n_cams > 1 ? (int)ev.cam[e] : 0) |
                                         ((long long)(in.stereo ? 1 : 0) << 16) | ((long long)(schur_act ? 1 : 0) << 17) |
                                         ((long long)(schur_act ? l : 0) << 32));
            double JL[3][3];
            jac_landmark(g.Xc, pose, in.cam, in.stereo, JL);
            lc = lm_contrib(JL, g, in.stereo);
            if (!(fl & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P)))
            {
                double JP[3][6];
                jac_pose(g.Xc, in.cam, in.stereo, JP);
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int r = 0; r < 6; r++)
                    {
                        // (the third Jacobian rows are zero for a monocular edge: the term is added
                        // unconditionally — a per-lane condition here is an exec-mask region per entry)
                        double s = JP[0][r] * JL[0][c] + JP[1][r] * JL[1][c];
                        s += JP[2][r] * JL[2][c];
                        H[c * 6 + r] = g.w * s;
                    }
            }
        }
    }
    {
        const int t = threadIdx.x;
        cs[0][t] = lc.h00, cs[1][t] = lc.h01, cs[2][t] = lc.h02, cs[3][t] = lc.h11, cs[4][t] = lc.h12;
        cs[5][t] = lc.h22, cs[6][t] = lc.b0, cs[7][t] = lc.b1, cs[8][t] = lc.b2;
#pragma unroll
        for (int k = 0; k < 8; k++)
            rs_[9 * t + k] = r8[k];
    }
    chi = block_sum(chi, sm); // contains the barrier that publishes cs[] and rs_[]
    if (threadIdx.x == 0)
        partials[blockIdx.x] = chi;
    { // records of this block's 256 slots -> global, fully coalesced (512 B per wave instruction)
        const long base = 8L * blockIdx.x * BS;
        const long limit = 8L * ev.E;
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            const int f = i * BS + threadIdx.x; // flat double index within the block's records
            if (base + f < limit)
                rec[base + f] = rs_[9 * (f >> 3) + (f & 7)];
        }
    }
    if (fuse)
        __syncthreads(); // the record area becomes ivs
    if (l >= 0 && l < ev.L && ev.lm_ptr[l] == e)
    { // owner of landmark l
        const int e1 = ev.lm_ptr[l + 1];
        const int bend = min(e1, (int)(blockIdx.x + 1) * BS);
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int ee = e; ee < bend; ee++)
        {
            const int t = ee - blockIdx.x * BS;
#pragma unroll
            for (int v = 0; v < 9; v++)
                a[v] += cs[v][t];
        }
        for (int ee = bend; ee < e1; ee++)
        { // edges of this landmark beyond the block (never with the engine's padded layout)
            const uint8_t fl = ev.flags[ee];
            if (fl & CUGO_EDGE_INACTIVE)
                continue;
            const EdgeIn in = load_edge(ev, ee, fl);
            const double* pose = poses + 7 * (size_t)in.ip;
            EdgeGeom g;
            edge_residual(pose, lms + 3 * (size_t)in.il, in.mu, in.mv, in.mr, in.stereo, in.omega,
                          in.cam, in.stereo ? rk.s : rk.m, g);
            double JL[3][3];
            jac_landmark(g.Xc, pose, in.cam, in.stereo, JL);
            const LmContrib c2 = lm_contrib(JL, g, in.stereo);
            a[0] += c2.h00, a[1] += c2.h01, a[2] += c2.h02, a[3] += c2.h11, a[4] += c2.h12;
            a[5] += c2.h22, a[6] += c2.b0, a[7] += c2.b1, a[8] += c2.b2;
        }
        double* Hg = Hll + 9 * (size_t)l;
        const double h9[9] = {a[0], a[1], a[2], a[1], a[3], a[4], a[2], a[4], a[5]};
#pragma unroll
        for (int i = 0; i < 9; i++)
            Hg[i] = h9[i];
        bl[3 * (size_t)l] = a[6], bl[3 * (size_t)l + 1] = a[7], bl[3 * (size_t)l + 2] = a[8];
        if (gform)
        {
            // Hll + lambda I = L L^T (3 x 3; a pivot that is not positive gives NaNs, the factorisation of Hsc
            // raises its flag on them and the trial is rejected: its retry takes the two-stream path), L^-1, y
            const double l00 = sqrt(a[0] + fuse_lambda);
            const double i00 = 1.0 / l00;
            const double l10 = a[1] * i00, l20 = a[2] * i00;
            const double l11 = sqrt(a[3] + fuse_lambda - l10 * l10);
            const double i11 = 1.0 / l11;
            const double l21 = (a[4] - l20 * l10) * i11;
            const double l22 = sqrt(a[5] + fuse_lambda - l20 * l20 - l21 * l21);
            const double i22 = 1.0 / l22;
            const double i10 = -(l10 * i00) * i11;
            const double i21 = -(l21 * i11) * i22;
            const double i20 = -(l20 * i00 + l21 * i10) * i22;
            double* q = ivs[threadIdx.x];
            q[0] = i00, q[1] = i10, q[2] = i11, q[3] = i20, q[4] = i21, q[5] = i22;
            double* lr = lmrec + 16 * (size_t)l; // one 128-byte line per landmark
            lr[0] = i00, lr[1] = i10, lr[2] = i11, lr[3] = i20, lr[4] = i21, lr[5] = i22;
            lr[6] = i00 * a[6];
            lr[7] = i10 * a[6] + i11 * a[7];
            lr[8] = i20 * a[6] + i21 * a[7] + i22 * a[8];
        }
        else if (fuse)
        {
            const Sym3 iv = sym3_inv(h9, fuse_lambda);
            double* o = invHll + 9 * (size_t)l;
            o[0] = iv.b00, o[1] = iv.b01, o[2] = iv.b02;
            o[3] = iv.b01, o[4] = iv.b11, o[5] = iv.b12;
            o[6] = iv.b02, o[7] = iv.b12, o[8] = iv.b22;
            double* q = ivs[threadIdx.x];
            q[0] = iv.b00, q[1] = iv.b01, q[2] = iv.b02, q[3] = iv.b11, q[4] = iv.b12, q[5] = iv.b22;
        }
    }
    if (e < ev.L && ev.lm_ptr[e] == ev.lm_ptr[e + 1])
    { // landmark without any edge
        double* H = Hll + 9 * (size_t)e;
#pragma unroll
        for (int i = 0; i < 9; i++)
            H[i] = 0;
        bl[3 * (size_t)e] = 0, bl[3 * (size_t)e + 1] = 0, bl[3 * (size_t)e + 2] = 0;
    }
    // ---- Hpl blocks of the block's 256 slots -> global through LDS: a lane storing its own
    // 144 bytes (9 stores at a 144-B stride) touches 64 cache lines per store instruction
    __syncthreads(); // cs is no longer read, ivs is complete
    bool act = false;
    double q[6] = {0, 0, 0, 0, 0, 0};
    if (fuse && T != nullptr && e < ev.E && l < ev.L)
    { // invHll of this slot's landmark, before the pool takes the Hpl blocks
        const int to = ev.lm_ptr[l] - (int)blockIdx.x * BS; // owner slot of the landmark (inside this block)
        if (to >= 0 && to < BS)
        {
#pragma unroll
            for (int i = 0; i < 6; i++)
                q[i] = ivs[to][i];
            act = !(ev.flags[e] & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE));
        }
    }
    if (fuse)
        __syncthreads(); // ivs has been read
    if (!gform)
    {
        double* mine = reinterpret_cast<double*>(pool2) + 18 * threadIdx.x;
#pragma unroll
        for (int i = 0; i < 18; i++)
            mine[i] = H[i];
    }
    __syncthreads();
    {
        const int ebase = blockIdx.x * BS;
        const long nvalid = 9L * max(0, min(BS, ev.E - ebase));
        if (!gform)
        {
#pragma unroll
            for (int i = 0; i < 9; i++)
            {
                const int idx = i * BS + threadIdx.x;
                if (idx < nvalid)
                {
                    const double2 v = pool2[idx];
                    st_pair(Hpl, 9 * (size_t)ebase + idx, v.x, v.y);
                }
            }
        }
        if (fuse && T != nullptr)
        { // T = Hpl invHll (or G = Hpl L^-T) of this block's slots, from the blocks still in registers
            if (!gform)
                __syncthreads(); // the Hpl blocks have left the pool
            double* mine = reinterpret_cast<double*>(pool2) + 18 * threadIdx.x;
#pragma unroll
            for (int r = 0; r < 6; r++)
            { // the stored Hpl values (float storage rounds them) are what k_schur_edges would read
                const double a = (double)(S)H[r], b = (double)(S)H[6 + r], c = (double)(S)H[12 + r];
                if (gform)
                { // q = L^-1: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2);  G[:, m] = sum_{n <= m} H[:, n] L^-1[m][n]
                    mine[r] = act ? H[r] * q[0] : 0.0;
                    mine[6 + r] = act ? H[r] * q[1] + H[6 + r] * q[2] : 0.0;
                    mine[12 + r] = act ? H[r] * q[3] + H[6 + r] * q[4] + H[12 + r] * q[5] : 0.0;
                    continue;
                }
                mine[r] = act ? a * q[0] + b * q[1] + c * q[2] : 0.0;
                mine[6 + r] = act ? a * q[1] + b * q[3] + c * q[4] : 0.0;
                mine[12 + r] = act ? a * q[2] + b * q[4] + c * q[5] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 9; i++)
            {
                const int idx = i * BS + threadIdx.x;
                if (idx < nvalid)
                {
                    const double2 v = pool2[idx];
                    st_pair(T, 9 * (size_t)ebase + idx, v.x, v.y);
                }
            }
        }
    }
}

// index of (r,c), r<=c in the packed upper triangle of a 6x6
__device__ __forceinline__ constexpr int tri6(int r, int c) { return r * 6 - r * (r - 1) / 2 + (c - r); }

// ---------------------------------------------------------------- build: poses ---------
// 32 accumulators per lane -> their 64-lane sums, in 32 shuffles: at every step a lane keeps one half of its live
// values and hands the other half to its partner (lane ^ 32, 16, 8, 4, 2), so the number of live values halves while
// the number of lanes summed doubles; a last exchange with lane ^ 1 completes the sum.  Afterwards a[0] of lane L
// holds the sum of accumulator L >> 1 over the whole wave.  Fixed order: bit-reproducible.  (A butterfly over all 32
// values would take 6 x 32 shuffles; through LDS the 27 x 256 values of a workgroup cost 57 KB of LDS per workgroup —
// two workgroups per CU, which is what bounded k_build_poses: latency, not arithmetic.)
__device__ __forceinline__ void wave_reduce32(double (&a)[32])
{
    const int lane = threadIdx.x & 63;
#define CUGO_HALVE(N, OFF)                                     \
    {                                                          \
        const bool hi = (lane & OFF) != 0;                     \
        _Pragma("unroll") for (int i = 0; i < N; i++)          \
        {                                                      \
            const double send = hi ? a[i] : a[i + N];          \
            const double keep = hi ? a[i + N] : a[i];          \
            a[i] = keep + __shfl_xor(send, OFF, 64);           \
        }                                                      \
    }
    CUGO_HALVE(16, 32)
    CUGO_HALVE(8, 16)
    CUGO_HALVE(4, 8)
    CUGO_HALVE(2, 4)
    CUGO_HALVE(1, 2)
#undef CUGO_HALVE
    a[0] += __shfl_xor(a[0], 1, 64);
}

// Hpp[p] = sum w JP^T JP, bp[p] = sum w JP^T e; one workgroup per pose, one lane per edge.
// The edge geometry comes from the 64-byte records k_build_edges left behind (Xc, e, w, camera
// index, stereo bit): each wave fetches the records of its 64 edges with 4 load instructions
// (4 lanes x 16 B per record: one cache line per edge instead of ~8 scattered lines for the
// planar measurement / index / landmark gathers), parks them in LDS and every lane reads its own.
// The 27 sums of a wave are formed in registers (wave_reduce32), the four waves' partial sums meet in 1 KB of LDS:
// 19 KB of LDS per workgroup, so that every pose of a kitti_00-sized graph is resident at once.
constexpr int BP_W = BS / 64;
__global__ __launch_bounds__(BS) void k_build_poses(EV ev, const double* __restrict__ rec,
                                                    double* __restrict__ Hpp,
                                                    double* __restrict__ bp)
{
    __shared__ double stage_all[BP_W][64 * 9]; // per wave: 64 records x 9 doubles
    __shared__ double wsum[BP_W][32];
    double* stage = stage_all[threadIdx.x >> 6];
    const int p = blockIdx.x;
    const int lane = threadIdx.x & 63;
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; i++)
        acc[i] = 0;
    const int i0 = ev.pose_ptr[p], i1 = ev.pose_ptr[p + 1];
    const double2* rec2 = reinterpret_cast<const double2*>(rec);
    // The work of a workgroup is a chain of dependent round trips (list entry -> record -> arithmetic), not
    // arithmetic: the entries of TWO rounds (this wave's 64 list entries [ibase, ibase + 64) and the 64 a workgroup
    // width further on) are fetched together, then the records of both, and only then are they consumed — a pose
    // with up to 512 edges (the kitti_00 shape: ~424) pays the chain once.
    auto fetch = [&](int e, double2 (&v)[4]) {
#pragma unroll
        for (int q = 0; q < 4; q++)
        { // 16 records per instruction: lanes 4j..4j+3 read the 4 quarters of record 16q + j
            const int j = 16 * q + (lane >> 2), part = lane & 3;
            const int ej = __shfl(e, j, 64);
            v[q] = rec2[4 * (size_t)ej + part];
        }
    };
    auto consume = [&](const double2 (&v)[4], int i) {
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int j = 16 * q + (lane >> 2), part = lane & 3;
            stage[9 * j + 2 * part] = v[q].x;
            stage[9 * j + 2 * part + 1] = v[q].y;
        }
        wave_sync_lds();
        if (i < i1)
        {
            const double* r = stage + 9 * lane;
            const double Xc[3] = {r[0], r[1], r[2]};
            const double ee[3] = {r[3], r[4], r[5]};
            const double w = r[6];
            const long long meta = __double_as_longlong(r[7]);
            const bool stereo = ((meta >> 16) & 1) != 0;
            const double* cam = ev.cams + 5 * (int)(meta & 0xFFFF);
            double JP[3][6];
            jac_pose(Xc, cam, stereo, JP);
            int k = 0;
#pragma unroll
            for (int rr = 0; rr < 6; rr++)
#pragma unroll
                for (int c = rr; c < 6; c++)
                {
                    double sacc = JP[0][rr] * JP[0][c] + JP[1][rr] * JP[1][c];
                    sacc += JP[2][rr] * JP[2][c]; // zero row for a monocular edge (see k_build_edges)
                    acc[k++] += w * sacc;
                }
#pragma unroll
            for (int rr = 0; rr < 6; rr++)
            {
                double sacc = JP[0][rr] * ee[0] + JP[1][rr] * ee[1];
                sacc += JP[2][rr] * ee[2];
                acc[21 + rr] += w * sacc;
            }
        }
        wave_sync_lds(); // the slot is refilled by the next round
    };
    for (int ibase = i0 + (threadIdx.x & ~63); ibase < i1; ibase += 2 * BS)
    { // wave-uniform loop
        const int ia = ibase + lane, ib = ia + BS;
        const bool second = ibase + BS < i1; // (wave-uniform)
        const int ea = ev.pose_edge[min(ia, i1 - 1)];
        const int eb = ev.pose_edge[min(ib, i1 - 1)];
        double2 va[4], vb[4];
        fetch(ea, va);
        fetch(eb, vb); // (unconditional: the clamped entries are valid, and a branch here would serialise the two chains)
        consume(va, ia);
        if (second)
            consume(vb, ib);
    }
    wave_reduce32(acc);
    if (!(lane & 1))
        wsum[threadIdx.x >> 6][lane >> 1] = acc[0];
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 42)
    {
        int k;
        if (t < 36)
        {
            const int r = t % 6, c = t / 6;
            k = tri6(r < c ? r : c, r < c ? c : r);
        }
        else
            k = 21 + (t - 36);
        double sum = wsum[0][k];
#pragma unroll
        for (int q = 1; q < BP_W; q++) // wave order
            sum += wsum[q][k];
        if (t < 36)
            Hpp[36 * (size_t)p + t] = sum;
        else
            bp[6 * (size_t)p + (t - 36)] = sum;
    }
}

// ---------------------------------------------------------------- fused iteration: pose pass
// From the second LM iteration on the build pass knows the damping of the first trial (launch_build: fuse_lambda) and
// leaves invHll and T.  Then the diagonal block of the Schur complement, bp and bsc of a pose need no pass of their
// own over the 144-byte T / Hpl blocks (k_hsc_diag*: two gathered blocks = four cache lines per edge) behind the pose
// pass (k_build_poses: one 64-byte record per edge): everything about an edge follows from that record and ONE more
// line of its landmark.  With Hpl_e = w JP^T JL (JL from the record's Xc and the rotation of the workgroup's own pose)
// and T_e = Hpl_e invHll_l,
//     w JP^T JP - T_e Hpl_e^T = JP^T N JP,   N = w I - w^2 JL invHll JL^T           (3 x 3, symmetric)
//     w JP^T e  - T_e bl_l    = JP^T u,      u = w (e - JL z),  z = invHll bl       (z comes with the landmark's line)
// so no 6 x 3 block is ever formed:  Hsc_pp = sum_e JP^T N JP,  bp = sum_e JP^T (w e),  bsc = sum_e JP^T u.
// k_build_edges leaves {invHll (6), z (3)} of landmark l in a 128-byte slot lmrec[16 l] and the landmark index and an
// "active" bit in the record.  One workgroup per pose, one lane per edge, eight waves: a pose of the kitti_00 shape
// (~424 edges) is ONE round — list entry -> record -> landmark line -> arithmetic -> sums.  Replaces k_build_poses +
// k_hsc_diag_mfma of that iteration; Hpp is NOT written (the engine makes up for it where a rejected trial needs it).
// (In the float mode the stored T / Hpl blocks are rounded to float and the off-diagonal blocks are formed from
// those; the diagonal blocks formed here never see a stored block and come out in full precision.)
constexpr int PS_W = 4, PS_BS = 64 * PS_W;
__global__ __launch_bounds__(PS_BS, 3) void k_pose_schur(EV ev, const double* __restrict__ rec,
                                                      const double* __restrict__ lmrec,
                                                      const double* __restrict__ poses,
                                                      const int32_t* __restrict__ rowptr, double lambda_diag,
                                                      double* __restrict__ Hsc, double* __restrict__ bp,
                                                      double* __restrict__ bsc)
{
    __shared__ double stage_all[PS_W][64 * 9]; // per wave: 64 records x 9 doubles
    __shared__ double wsum[PS_W][34];
    const int p = xcd_contiguous_item(ev.P);
    if (p >= ev.P)
        return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double* stage = stage_all[w];
    // the rotation of the pose: formed once, parked in LDS and re-read by every round (nine values per lane that would
    // otherwise sit in registers for the whole kernel, next to 33 accumulators)
    __shared__ double Rs[9];
    if (threadIdx.x == 0)
    {
        double R0[3][3];
        quat_to_rot(poses + 7 * (size_t)p, R0);
#pragma unroll
        for (int a = 0; a < 9; a++)
            Rs[a] = R0[a / 3][a % 3];
    }
    __syncthreads();
    double acc[32], acc32 = 0;
#pragma unroll
    for (int i = 0; i < 32; i++)
        acc[i] = 0;
    const int i0 = ev.pose_ptr[p], i1 = ev.pose_ptr[p + 1];
    const double2* rec2 = reinterpret_cast<const double2*>(rec);
    for (int ibase = i0 + 64 * w; ibase < i1; ibase += PS_BS)
    { // wave-uniform loop
        const int i = ibase + lane;
        const int e = ev.pose_edge[min(i, i1 - 1)];
        double2 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
        { // 16 records per instruction: lanes 4j..4j+3 read the 4 quarters of record 16q + j
            const int j = 16 * q + (lane >> 2), part = lane & 3;
            const int ej = __shfl(e, j, 64);
            v[q] = rec2[4 * (size_t)ej + part];
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int j = 16 * q + (lane >> 2), part = lane & 3;
            stage[9 * j + 2 * part] = v[q].x;
            stage[9 * j + 2 * part + 1] = v[q].y;
        }
        wave_sync_lds();
        const double* r = stage + 9 * lane;
        const double Xc[3] = {r[0], r[1], r[2]};
        const double wgt = i < i1 ? r[6] : 0.0;
        const double we[3] = {wgt * r[3], wgt * r[4], wgt * r[5]};
        const long long meta = __double_as_longlong(r[7]);
        wave_sync_lds(); // (the slot is refilled by the next round)
        const bool stereo = ((meta >> 16) & 1) != 0;
        const bool act = i < i1 && ((meta >> 17) & 1) != 0;
        const double* cam = ev.cams + 5 * (int)(meta & 0xFFFF);
        // the landmark's line (an inactive edge reads slot 0 and never uses it).  (Measured: a pose-major copy of the
        // landmark indices, so that this line is requested together with the record instead of behind it, changes
        // nothing — 34.5 vs 32.2 us: the kernel is bound by its 168 registers per edge, not by this chain.)
        const double2* lr = reinterpret_cast<const double2*>(lmrec + 16 * (size_t)(act ? (int)(meta >> 32) : 0));
        const double2 q01 = lr[0], q23 = lr[1], q45 = lr[2], z01 = lr[3]; // L^-1: (0,0) (1,0) | (1,1) (2,0) | (2,1) (2,2); y
        const double z2 = lr[4].x;
        double N[6], u[3]; // N: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
        {
            int zero = 0;
            asm volatile("" : "+v"(zero)); // (keeps the nine reads inside the loop)
            double R[3][3];
#pragma unroll
            for (int a = 0; a < 9; a++)
                R[a / 3][a % 3] = Rs[a + zero];
            double JL[3][3];
            jac_landmark_R(Xc, R, cam, stereo, JL);
            double K[3][3]; // A = JL L^-T:  A[m][j] = sum_{i <= j} JL[m][i] L^-1[j][i]
#pragma unroll
            for (int m = 0; m < 3; m++)
            {
                K[m][0] = JL[m][0] * q01.x;
                K[m][1] = JL[m][0] * q01.y + JL[m][1] * q23.x;
                K[m][2] = JL[m][0] * q23.y + JL[m][1] * q45.x + JL[m][2] * q45.y;
            }
            // (an inactive edge has read slot 0 of the landmark lines — memory nobody may have written when no landmark
            // is free: its terms are SELECTED away, 0 x whatever sits there would not do)
            const double w2 = wgt * wgt;
            int k = 0;
#pragma unroll
            for (int m = 0; m < 3; m++)
#pragma unroll
                for (int n = m; n < 3; n++)
                {
                    const double mm = K[m][0] * K[n][0] + K[m][1] * K[n][1] + K[m][2] * K[n][2]; // (A A^T)[m][n]
                    N[k++] = (m == n ? wgt : 0.0) - (act ? w2 * mm : 0.0);
                }
#pragma unroll
            for (int m = 0; m < 3; m++)
            {
                const double jz = K[m][0] * z01.x + K[m][1] * z01.y + K[m][2] * z2; // JL z = A y
                u[m] = we[m] - (act ? wgt * jz : 0.0);
            }
        }
        __builtin_amdgcn_sched_barrier(0); // (JL, K and the landmark's line are dead before JP exists: registers)
        double JP[3][6];
        jac_pose(Xc, cam, stereo, JP);
        // (the right-hand sides first: we and u are dead before the 21 block entries are formed)
#pragma unroll
        for (int rr = 0; rr < 6; rr++)
        {
            acc[21 + rr] += JP[0][rr] * we[0] + JP[1][rr] * we[1] + JP[2][rr] * we[2];
            const double bs = JP[0][rr] * u[0] + JP[1][rr] * u[1] + JP[2][rr] * u[2];
            if (rr < 5)
                acc[27 + rr] += bs;
            else
                acc32 += bs;
        }
        __builtin_amdgcn_sched_barrier(0);
        int k = 0;
#pragma unroll
        for (int c = 0; c < 6; c++)
        { // column c of G = N JP, then the entries (r <= c) of JP^T G (accumulator c (c + 1) / 2 + r)
            const double g0 = N[0] * JP[0][c] + N[1] * JP[1][c] + N[2] * JP[2][c];
            const double g1 = N[1] * JP[0][c] + N[3] * JP[1][c] + N[4] * JP[2][c];
            const double g2 = N[2] * JP[0][c] + N[4] * JP[1][c] + N[5] * JP[2][c];
#pragma unroll
            for (int rr = 0; rr <= c; rr++)
                acc[k++] += JP[0][rr] * g0 + JP[1][rr] * g1 + JP[2][rr] * g2;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wave_reduce32(acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        acc32 += __shfl_xor(acc32, off, 64);
    if (!(lane & 1))
        wsum[w][lane >> 1] = acc[0];
    if (lane == 0)
        wsum[w][32] = acc32;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 48)
    {
        int k;
        if (t < 36)
        { // accumulator of (r <= c): c (c + 1) / 2 + r
            const int rr = t % 6, c = t / 6;
            const int lo = rr < c ? rr : c, hi = rr < c ? c : rr;
            k = hi * (hi + 1) / 2 + lo;
        }
        else
            k = 21 + (t - 36);
        double sum = wsum[0][k];
#pragma unroll
        for (int q = 1; q < PS_W; q++) // wave order
            sum += wsum[q][k];
        if (t < 36)
            Hsc[36 * (size_t)rowptr[p] + t] = (t % 6 == t / 6) ? sum + lambda_diag : sum;
        else if (t < 42)
            bp[6 * (size_t)p + (t - 36)] = sum;
        else
            bsc[6 * (size_t)p + (t - 42)] = sum;
    }
}


// ---------------------------------------------------------------- max diagonal ---------
__global__ __launch_bounds__(BS) void k_max_diag(const double* __restrict__ Hpp, int nP,
                                                 const double* __restrict__ Hll, int nL,
                                                 double* __restrict__ partials)
{
    __shared__ double sm[BS];
    const long n = 6L * nP + 3L * nL;
    double v = 0;
    for (long i = (long)blockIdx.x * BS + threadIdx.x; i < n; i += (long)gridDim.x * BS)
    {
        if (i < 6L * nP)
        {
            const long j = i / 6, k = i % 6;
            v = fmax(v, Hpp[36 * j + 7 * k]);
        }
        else
        {
            const long ii = i - 6L * nP;
            const long j = ii / 3, k = ii % 3;
            v = fmax(v, Hll[9 * j + 4 * k]);
        }
    }
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int s = BS / 2; s > 0; s >>= 1)
    {
        if ((int)threadIdx.x < s)
            sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        partials[blockIdx.x] = sm[0];
}

// ---------------------------------------------------------------- Schur: edges ---------
// invHll = (Hll + lambda I)^-1 (written by the landmark's first edge lane),
// T[e] = Hpl[e] * invHll   (ref: computeBschureKernel .cu:1286-1314, lane per edge)
// The 256 Hpl blocks of a workgroup (36 KB, contiguous) go through LDS: coalesced 16-B loads in,
// lane e works on its own 18 values in place, coalesced stores of T out.  A lane reading its
// block straight from memory (9 loads at a 144-B stride) makes every load instruction touch 64
// cache lines; the vector memory path, not HBM, then sets the pace.
template <typename S>
__global__ __launch_bounds__(BS) void k_schur_edges(EV ev, double lambda,
                                                    const double* __restrict__ Hll,
                                                    const S* __restrict__ Hpl,
                                                    double* __restrict__ invHll,
                                                    S* __restrict__ T)
{
    __shared__ double2 hs[BS * 9 + 1];
    const int t = threadIdx.x;
    const int ebase = blockIdx.x * BS;
    const int e = ebase + t;
    const long nvalid = 9L * max(0, min(BS, ev.E - ebase)); // pairs of this block that exist
    {
        double2 v[9];
#pragma unroll
        for (int i = 0; i < 9; i++)
        {
            const int idx = i * BS + t;
            v[i] = ld_pair(Hpl, 9 * (size_t)ebase + (size_t)min((long)idx, max(nvalid - 1, 0L)));
        }
#pragma unroll
        for (int i = 0; i < 9; i++)
            hs[i * BS + t] = v[i];
    }
    bool act = false;
    Sym3 iv = {0, 0, 0, 0, 0, 0};
    if (e < ev.E)
    {
        const int l = ev.lm[e];
        if (l < ev.L) // fixed landmark: no Hll block
        {
            iv = sym3_inv(Hll + 9 * (size_t)l, lambda);
            if (e == ev.lm_ptr[l])
            {
                double* o = invHll + 9 * (size_t)l;
                o[0] = iv.b00, o[1] = iv.b01, o[2] = iv.b02;
                o[3] = iv.b01, o[4] = iv.b11, o[5] = iv.b12;
                o[6] = iv.b02, o[7] = iv.b12, o[8] = iv.b22;
            }
            act = !(ev.flags[e] & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE));
        }
    }
    __syncthreads();
    {
        double* H = reinterpret_cast<double*>(hs) + 18 * t; // this lane's block, overwritten by T
        double Tt[18];
#pragma unroll
        for (int r = 0; r < 6; r++)
        {
            const double a = H[r], b = H[6 + r], c = H[12 + r];
            Tt[r] = a * iv.b00 + b * iv.b01 + c * iv.b02;
            Tt[6 + r] = a * iv.b01 + b * iv.b11 + c * iv.b12;
            Tt[12 + r] = a * iv.b02 + b * iv.b12 + c * iv.b22;
        }
        // edges without a T block (fixed endpoint, padding) store zeros: nothing reads them
#pragma unroll
        for (int i = 0; i < 18; i++)
            H[i] = act ? Tt[i] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        const int idx = i * BS + t;
        if (idx < nvalid)
        {
            const double2 v = hs[idx];
            st_pair(T, 9 * (size_t)ebase + idx, v.x, v.y);
        }
    }
}

// ---------------------------------------------------------------- Schur: landmark-major -
// k_schur_fused = k_schur_edges + the H-side and b-side products of the group's own landmarks.
// (ref: computeBschureKernel + computeHschureKernel, src/cuda/cuda_block_solver.cu:1286-1345.)
// The 256 Hpl blocks of the workgroup are in LDS anyway (36 KB); T = Hpl invHll is kept beside
// them (36 KB) and every product T_a Hpl_b^T of the group's landmarks is formed from LDS: each Hpl
// block is read from HBM ONCE for the whole Schur complement and T never has to be written (it
// is, if the caller passes a T array: the C ABI's d_T output).
// Products are summed per (group, destination block) — a "partial slot" of the host-built plan
// (csrc/host/schur_plan.cpp) — by ONE group of 6 lanes (lane = row r of the 6x6 sum, 6 column
// accumulators in registers), in list order: no atomics, fixed order.  A wave runs 10 such lane
// groups at once.  Operand traffic per product and lane: 3 ds_read_b64 (its row of T_a) + 9
// ds_read_b128 (Hpl_b, the same address for the 6 lanes: broadcast) for 18 FMAs.
// k_hsc_reduce then adds the slots of every Hsc block in group order.
constexpr int SF_LG = 10;        // lane groups per wave
constexpr int SF_PCAP = 3072;    // products of a group staged in LDS (6 KB); longer lists are read from memory
constexpr int SF_SCAP = 640;     // partial slots of a group whose list bounds are staged in LDS
struct SchurPlanDev
{
    const int32_t* grp_ptr;
    const int32_t* grp_nwave;
    const int32_t* slot_rhs;
    const int32_t* slot_ptr;
    const uint16_t* prod;
    double* part_H;
    double* part_b;
};

// operands of one product for lane r of a lane group: its row of T_a and the whole Hpl_b
struct SfOperands
{
    double t0, t1, t2;
    double2 h[9];
};
__device__ __forceinline__ void sf_load(SfOperands& o, unsigned ab, int r, const double* __restrict__ hs,
                                        const double* __restrict__ ts)
{
    const int a = ab & 255, b = ab >> 8;
    const double* T = ts + 18 * a + r;
    o.t0 = T[0], o.t1 = T[6], o.t2 = T[12];
    const double2* H = reinterpret_cast<const double2*>(hs + 18 * b); // 144-B blocks: 16-B aligned
#pragma unroll
    for (int q = 0; q < 9; q++)
        o.h[q] = H[q];
}
__device__ __forceinline__ void sf_fma(const SfOperands& o, double (&acc)[6])
{
    const double* hd = reinterpret_cast<const double*>(o.h);
#pragma unroll
    for (int c = 0; c < 6; c++)
    { // (T_a Hpl_b^T)(r, c) = sum_m T_a(r, m) Hpl_b(c, m); blocks are 6x3 column-major
        double sacc = o.t0 * hd[c];
        sacc = fma(o.t1, hd[6 + c], sacc);
        sacc = fma(o.t2, hd[12 + c], sacc);
        acc[c] += sacc;
    }
}

// Products i0, i0 + stride, ... < i1 of a slot, summed in that order.  Software pipeline, two
// register sets in turn: the operands of the next product are read from LDS while the 18 FMAs of
// the current one run (a lane group works through its list alone: nothing else hides the latency).
// RHS: the slot is a diagonal block (a == b): also (T_a bl)(r).
template <bool RHS>
__device__ __forceinline__ void sf_slot_products(const uint16_t* __restrict__ pl, int i0, int i1, int stride, int r,
                                                 const double* __restrict__ hs, const double* __restrict__ ts,
                                                 const double* __restrict__ sbl, double (&acc)[6], double& rhs)
{
    if (i0 >= i1)
        return;
    SfOperands A, B;
    // list entries are read two products ahead, operands one ahead: neither LDS latency is exposed
    const int last = i1 - 1;
    unsigned abA = pl[i0], abB = pl[min(i0 + stride, last)], abC = pl[min(i0 + 2 * stride, last)];
    sf_load(A, abA, r, hs, ts);
    int i = i0;
    while (true)
    {
        // A holds product i, abB its successor's entry
        const int in = i + stride;
        const unsigned abD = pl[min(i + 3 * stride, last)];
        if (in < i1)
            sf_load(B, abB, r, hs, ts);
        double l0 = 0, l1 = 0, l2 = 0;
        if (RHS)
        {
            const double* bl3 = sbl + 3 * (abA & 255);
            l0 = bl3[0], l1 = bl3[1], l2 = bl3[2];
        }
        sf_fma(A, acc);
        if (RHS)
            rhs += A.t0 * l0 + A.t1 * l1 + A.t2 * l2;
        if (in >= i1)
            break;
        // B holds product in, abC the entry after it
        const int in2 = in + stride;
        const unsigned abE = pl[min(i + 4 * stride, last)];
        if (in2 < i1)
            sf_load(A, abC, r, hs, ts);
        if (RHS)
        {
            const double* bl3 = sbl + 3 * (abB & 255);
            l0 = bl3[0], l1 = bl3[1], l2 = bl3[2];
        }
        sf_fma(B, acc);
        if (RHS)
            rhs += B.t0 * l0 + B.t1 * l1 + B.t2 * l2;
        if (in2 >= i1)
            break;
        abA = abC, abB = abD, abC = abE;
        i = in2;
    }
}

constexpr int SF_BS_FWD = 1024;
// The product phase of k_schur_fused.  STAGED (compile time): the group's product list and slot
// bounds are in LDS; else they are read from memory.  One pointer that may be either would make
// every list access a flat one (waits on both memory counters, i.e. also on the partial-sum stores
// in flight).
template <bool STAGED>
__device__ __forceinline__ void sf_products(const SchurPlanDev& pl, int t, int s0, int s1, int p0,
                                            const uint16_t* __restrict__ sprod, const int32_t* __restrict__ sptr,
                                            const int32_t* __restrict__ srhs, const double* __restrict__ hs,
                                            const double* __restrict__ ts, const double* __restrict__ sbl,
                                            double (*wsc)[10 * 6 * 7], int g)
{
    // ---- products.  The group's slots come longest first (host plan).  The first n_wave of them
    // (more than SF_LONG products: the diagonal blocks, mostly) are taken by a whole wave each: lane
    // group lg forms every 10th product, the ten sums are added in lane-group order through LDS.
    // The others go to single lane groups: s_w + u, s_w + u + 80, ...
    const int lane = t & 63, w = t >> 6;
    const int lg = lane / 6, r = lane - 6 * lg;
    if (lg >= SF_LG)
        return;
    const int nwave = pl.grp_nwave[g];
    for (int sl = s0 + w; sl < s0 + nwave; sl += SF_BS_FWD / 64)
    {
        const int i0 = STAGED ? sptr[sl - s0] : pl.slot_ptr[sl] - p0;
        const int i1 = STAGED ? sptr[sl - s0 + 1] : pl.slot_ptr[sl + 1] - p0;
        const int ri = STAGED ? srhs[sl - s0] : pl.slot_rhs[sl];
        double acc[6] = {0, 0, 0, 0, 0, 0}, rhs = 0;
        const uint16_t* plist = STAGED ? sprod : pl.prod + p0;
        if (ri >= 0) // wave-uniform: the whole wave works on this slot
            sf_slot_products<true>(plist, i0 + lg, i1, SF_LG, r, hs, ts, sbl, acc, rhs);
        else
            sf_slot_products<false>(plist, i0 + lg, i1, SF_LG, r, hs, ts, sbl, acc, rhs);
        double* sc = wsc[w] + (6 * lg + r) * 7;
#pragma unroll
        for (int c = 0; c < 6; c++)
            sc[c] = acc[c];
        sc[6] = rhs;
        // wavefront scope: a wave's LDS operations complete in order, only the compiler must not
        // move them (a workgroup-scope fence would also wait for the partial stores in flight)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lg == 0)
        {
            double tot[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int g2 = 0; g2 < SF_LG; g2++)
#pragma unroll
                for (int c = 0; c < 7; c++)
                    tot[c] += wsc[w][(6 * g2 + r) * 7 + c];
            double* o = pl.part_H + 36 * (size_t)sl;
#pragma unroll
            for (int c = 0; c < 6; c++)
                o[6 * c + r] = tot[c];
            if (ri >= 0)
                pl.part_b[6 * (size_t)ri + r] = tot[6];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const int unit = w * SF_LG + lg;
    for (int sl = s0 + nwave + unit; sl < s1; sl += (SF_BS_FWD / 64) * SF_LG)
    {
        const int i0 = STAGED ? sptr[sl - s0] : pl.slot_ptr[sl] - p0;
        const int i1 = STAGED ? sptr[sl - s0 + 1] : pl.slot_ptr[sl + 1] - p0;
        const int ri = STAGED ? srhs[sl - s0] : pl.slot_rhs[sl];
        double acc[6] = {0, 0, 0, 0, 0, 0}, rhs = 0;
        const uint16_t* plist = STAGED ? sprod : pl.prod + p0;
        if (ri >= 0)
            sf_slot_products<true>(plist, i0, i1, 1, r, hs, ts, sbl, acc, rhs);
        else
            sf_slot_products<false>(plist, i0, i1, 1, r, hs, ts, sbl, acc, rhs);
        double* o = pl.part_H + 36 * (size_t)sl;
#pragma unroll
        for (int c = 0; c < 6; c++)
            o[6 * c + r] = acc[c];
        if (ri >= 0)
            pl.part_b[6 * (size_t)ri + r] = rhs;
    }
}

constexpr int SF_BS = 1024; // 16 waves on a 256-edge group (its LDS footprint allows one workgroup per CU anyway)
constexpr int SF_LR = (9 * BS + SF_BS - 1) / SF_BS; // load rounds
template <typename S>
__global__ __launch_bounds__(SF_BS) void k_schur_fused(EV ev, double lambda, const double* __restrict__ Hll,
                                                    const double* __restrict__ bl, const S* __restrict__ Hpl,
                                                    double* __restrict__ invHll, S* __restrict__ T,
                                                    SchurPlanDev pl)
{
    __shared__ double2 hs2[BS * 9];
    __shared__ double2 ts2[BS * 9];
    __shared__ double sbl[BS * 3];
    __shared__ uint16_t sprod[SF_PCAP];
    __shared__ int32_t sptr[SF_SCAP + 1], srhs[SF_SCAP];
    __shared__ double wsc[SF_BS / 64][SF_LG * 6 * 7]; // per wave: the lane groups' sums of a long slot
    const int t = threadIdx.x;
    const int g = blockIdx.x, ebase = g * BS;
    const int e = ebase + t;
    const long nvalid = 9L * max(0, min(BS, ev.E - ebase)); // pairs of this block that exist
    {
        double2 v[SF_LR];
#pragma unroll
        for (int i = 0; i < SF_LR; i++)
        {
            const int idx = i * SF_BS + t; // 2304 pairs of the group's 256 blocks
            v[i] = ld_pair(Hpl, 9 * (size_t)ebase + (size_t)min((long)min(idx, 9 * BS - 1), max(nvalid - 1, 0L)));
        }
#pragma unroll
        for (int i = 0; i < SF_LR; i++)
            if (i * SF_BS + t < 9 * BS)
                hs2[i * SF_BS + t] = v[i];
    }
    const int s0 = pl.grp_ptr[g], s1 = pl.grp_ptr[g + 1];
    const int p0 = pl.slot_ptr[s0], np = pl.slot_ptr[s1] - p0;
    const bool staged = np <= SF_PCAP && s1 - s0 <= SF_SCAP;
    if (staged)
    {
        for (int i = t; i < np; i += SF_BS)
            sprod[i] = pl.prod[p0 + i];
        for (int i = t; i <= s1 - s0; i += SF_BS)
            sptr[i] = pl.slot_ptr[s0 + i] - p0;
        for (int i = t; i < s1 - s0; i += SF_BS)
            srhs[i] = pl.slot_rhs[s0 + i];
    }
    bool act = false;
    Sym3 iv = {0, 0, 0, 0, 0, 0};
    double b0 = 0, b1 = 0, b2 = 0;
    if (t < BS && e < ev.E)
    {
        const int l = ev.lm[e];
        if (l < ev.L) // fixed landmark: no Hll block
        {
            iv = sym3_inv(Hll + 9 * (size_t)l, lambda);
            b0 = bl[3 * (size_t)l], b1 = bl[3 * (size_t)l + 1], b2 = bl[3 * (size_t)l + 2];
            if (e == ev.lm_ptr[l])
            {
                double* o = invHll + 9 * (size_t)l;
                o[0] = iv.b00, o[1] = iv.b01, o[2] = iv.b02;
                o[3] = iv.b01, o[4] = iv.b11, o[5] = iv.b12;
                o[6] = iv.b02, o[7] = iv.b12, o[8] = iv.b22;
            }
            act = !(ev.flags[e] & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE));
        }
    }
    if (t < BS)
        sbl[3 * t] = b0, sbl[3 * t + 1] = b1, sbl[3 * t + 2] = b2;
    __syncthreads();
    double* hs = reinterpret_cast<double*>(hs2);
    double* ts = reinterpret_cast<double*>(ts2);
    if (t < BS)
    {
        const double* H = hs + 18 * t; // this lane's block
        double* Tt = ts + 18 * t;
#pragma unroll
        for (int r = 0; r < 6; r++)
        {
            const double a = H[r], b = H[6 + r], c = H[12 + r];
            // edges without a T block (fixed endpoint, padding) hold zeros: no product refers to them
            Tt[r] = act ? a * iv.b00 + b * iv.b01 + c * iv.b02 : 0.0;
            Tt[6 + r] = act ? a * iv.b01 + b * iv.b11 + c * iv.b12 : 0.0;
            Tt[12 + r] = act ? a * iv.b02 + b * iv.b12 + c * iv.b22 : 0.0;
        }
    }
    __syncthreads();
    if (T)
    { // the reference's Hpl_invHll output, on request
#pragma unroll
        for (int i = 0; i < SF_LR; i++)
        {
            const int idx = i * SF_BS + t;
            if (idx < nvalid)
            {
                const double2 v = ts2[idx];
                st_pair(T, 9 * (size_t)ebase + idx, v.x, v.y);
            }
        }
    }
    if (staged)
        sf_products<true>(pl, t, s0, s1, p0, sprod, sptr, srhs, hs, ts, sbl, wsc, g);
    else
        sf_products<false>(pl, t, s0, s1, p0, sprod, sptr, srhs, hs, ts, sbl, wsc, g);
}

// Hsc[k] = [Hpp(p) (+ lambda I)] - sum of the block's partial slots (group order);
// bsc[p] = bp[p] - sum of the rhs partials of the diagonal block.  One wave per Hsc block, lane =
// element: a slot is 288 contiguous bytes.
__global__ __launch_bounds__(BS) void k_hsc_reduce(int nblocks, const int32_t* __restrict__ red_ptr,
                                                   const int32_t* __restrict__ red_slot,
                                                   const int32_t* __restrict__ slot_rhs,
                                                   const int32_t* __restrict__ blk_pose,
                                                   const double* __restrict__ part_H,
                                                   const double* __restrict__ part_b, double lambda_diag,
                                                   const double* __restrict__ Hpp, const double* __restrict__ bp,
                                                   double* __restrict__ Hsc, double* __restrict__ bsc)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = blockIdx.x * (BS / 64) + w;
    if (k >= nblocks)
        return;
    const int p = blk_pose[k]; // (all 64 lanes stay: each holds a slot id for the shuffles below)
    const int q0 = red_ptr[k], q1 = red_ptr[k + 1];
    const int el = lane < 36 ? lane : min(lane - 36, 5); // element of H (lanes 0..35) or of the rhs (36..41)
    double sum = 0;
    for (int qb = q0; qb < q1; qb += 64)
    { // the slot ids of up to 64 slots with one coalesced load, then eight partial loads in flight
        const int n = min(64, q1 - qb);
        const int mine = red_slot[qb + min(lane, n - 1)];
        const int mine_rhs = (p >= 0) ? slot_rhs[mine] : 0;
        for (int u0 = 0; u0 < n; u0 += 8)
        {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
            {
                const int src = min(u0 + u, n - 1);
                const int sl = __shfl(mine, src), rh = __shfl(mine_rhs, src);
                v[u] = lane < 36 ? part_H[36 * (size_t)sl + el] : (p >= 0 ? part_b[6 * (size_t)rh + el] : 0.0);
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (u0 + u < n)
                    sum += v[u];
        }
    }
    if (lane < 36)
    {
        double val = -sum;
        if (p >= 0)
        {
            val += Hpp[36 * (size_t)p + lane];
            if (lane % 7 == 0) // diagonal element of the 6x6 block
                val += lambda_diag;
        }
        Hsc[36 * (size_t)k + lane] = val;
    }
    else if (p >= 0 && lane < 42)
        bsc[6 * (size_t)p + el] = bp[6 * (size_t)p + el] - sum;
}

// ---------------------------------------------------------------- Schur: diagonal ------
// Hsc(p,p) = Hpp[p] (+lambda I) - sum_e T_e Hpl_e^T ; bsc[p] = bp[p] - sum_e T_e bl[l(e)]
// One workgroup per pose, its edge list dealt to the four waves in chunks of 7 edges.  As in
// k_hsc_offdiag the 144-B operands of a chunk are fetched by one pair of 16-B loads (lanes
// 9j..9j+8 read edge j's T and Hpl blocks: ~2 cache lines per block instead of 64 lines per
// instruction), staged in a per-wave LDS slot, and LANE = OUTPUT ELEMENT: lanes 0..20 own the
// 21 upper-triangle entries of the 6x6 sum, lanes 21..26 the 6 entries of the rhs sum.  The
// per-wave partial sums are added in wave order: fixed order, bit-reproducible.
typedef double hsc_d4 __attribute__((ext_vector_type(4)));
constexpr int HD_CH = 7;
constexpr int HD_BS = 1024; // 16 waves per pose: the per-wave chain of dependent loads is what costs
constexpr int HD_W = HD_BS / 64;
__device__ __forceinline__ void wave_sync_lds0()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
template <typename S>
__global__ __launch_bounds__(HD_BS) void k_hsc_diag(EV ev, const int32_t* __restrict__ rowptr,
                                                 double lambda_diag,
                                                 const double* __restrict__ Hpp,
                                                 const double* __restrict__ bp,
                                                 const double* __restrict__ bl,
                                                 const S* __restrict__ Hpl,
                                                 const S* __restrict__ T,
                                                 double* __restrict__ Hsc,
                                                 double* __restrict__ bsc)
{
    __shared__ double2 sT2[HD_W][HD_CH * 9 + 1]; // T blocks as loaded
    __shared__ double sU[HD_W][HD_CH * 21 + 3];  // per edge: 3 rows of [H[6m..6m+5], bl[m]]
    __shared__ double part[HD_W][28];
    const int p = xcd_contiguous_item(ev.P);
    if (p >= ev.P)
        return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // output element of this lane: v < 21 -> (r, c) of the upper triangle, else rhs row r
    const int v = lane < 27 ? lane : 26;
    int r = 0, cc = 6;
    if (v < 21)
    {
        int k = v;
        while (k >= 6 - r)
        {
            k -= 6 - r;
            r++;
        }
        cc = r + k;
    }
    else
        r = v - 21;
    const int pj = min(lane / 9, HD_CH - 1), part9 = lane - 9 * (lane / 9);
    const bool writer = lane < 9 * HD_CH;
    const double* sT = reinterpret_cast<const double*>(sT2[w]);
    double* su = sU[w];
    const int i0 = ev.pose_ptr[p], i1 = ev.pose_ptr[p + 1];
    const int nch = (i1 - i0 + HD_CH - 1) / HD_CH;
    double acc = 0;
    int e_next = 0;
    if (w < nch)
        e_next = ev.pose_edge[min(i0 + HD_CH * w + pj, i1 - 1)];
    for (int ch = w; ch < nch; ch += HD_W)
    {
        const int i = i0 + HD_CH * ch + pj;
        const int e = e_next;
        if (ch + HD_W < nch)
            e_next = ev.pose_edge[min(i + HD_CH * HD_W, i1 - 1)];
        const uint8_t fl = ev.flags[e];
        const int l = ev.lm[e];
        const bool act = i < i1 && !(fl & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE));
        // fixed-landmark edges have l >= L: no bl entry (and they are never active)
        const int lsafe = act ? l : 0;
        double2 tv = ld_pair(T, 9 * (size_t)e + part9);
        const double2 hv = ld_pair(Hpl, 9 * (size_t)e + part9);
        const double bvl = bl[3 * (size_t)lsafe + min(part9, 2)];
        const double bv = act ? bvl : 0.0; // (see k_hsc_diag_mfma)
        if (!act)
            tv = make_double2(0, 0); // the edge contributes nothing
        if (writer)
        {
            sT2[w][9 * pj + part9] = tv;
            const int k0 = 2 * part9, k1 = k0 + 1;
            su[21 * pj + 7 * (k0 / 6) + k0 % 6] = hv.x;
            su[21 * pj + 7 * (k1 / 6) + k1 % 6] = hv.y;
            if (part9 < 3)
                su[21 * pj + 7 * part9 + 6] = bv;
        }
        wave_sync_lds0();
        const int n = min(HD_CH, i1 - i0 - HD_CH * ch);
#pragma unroll
        for (int u = 0; u < HD_CH; u++)
            if (u < n) // wave-uniform
                acc += sT[18 * u + r] * su[21 * u + cc] + sT[18 * u + 6 + r] * su[21 * u + 7 + cc] +
                       sT[18 * u + 12 + r] * su[21 * u + 14 + cc];
        wave_sync_lds0(); // the slot is rewritten by the next chunk
    }
    if (lane < 27)
        part[w][lane] = acc;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 36)
    {
        const int rr = t % 6, c = t / 6;
        const int a = rr < c ? rr : c, b = rr < c ? c : rr;
        const int k = tri6(a, b);
        double sum = 0;
        for (int q = 0; q < HD_W; q++)
            sum += part[q][k];
        double val = Hpp[36 * (size_t)p + t] - sum;
        if (rr == c)
            val += lambda_diag;
        Hsc[36 * (size_t)rowptr[p] + t] = val;
    }
    else if (t < 42)
    {
        const int k = 21 + (t - 36);
        double sum = 0;
        for (int q = 0; q < HD_W; q++)
            sum += part[q][k];
        bsc[6 * (size_t)p + (t - 36)] = bp[6 * (size_t)p + (t - 36)] - sum;
    }
}

// The same sums on the matrix cores (see k_hsc_offdiag_mfma for the why: the vector form is bound by the LDS
// port, six 8-byte reads per edge and lane).  A wave takes 14 edges per chunk: A = the T blocks (group 0 in tile
// rows 0..5, group 1 in rows 6..11), B = per K column [H(0..5, kk) | bl[kk]]: tile columns 0..5 / 6..11 the
// Hpl blocks of the two groups, columns 12 / 13 their bl entries.  Result of a wave:
// D[0:6,0:6] + D[6:12,6:12] and D[0:6,12] + D[6:12,13]; the waves' partial sums are added in wave order.
constexpr int HM_BS = 1024, HM_W = HM_BS / 64, HM_CH = 14;
constexpr int HM_ZT = 252, HM_ZU = 294; // a zero behind the staged chunk (doubles)
template <typename S>
__global__ __launch_bounds__(HM_BS, 8) void k_hsc_diag_mfma(EV ev, const int32_t* __restrict__ rowptr,
                                                      double lambda_diag,
                                                      const double* __restrict__ Hpp,
                                                      const double* __restrict__ bp,
                                                      const double* __restrict__ bl,
                                                      const S* __restrict__ Hpl,
                                                      const S* __restrict__ T,
                                                      double* __restrict__ Hsc,
                                                      double* __restrict__ bsc)
{
    __shared__ double2 sT2[HM_W][128];  // T blocks as loaded: group g at double2 63 g; later the wave's D tile
    __shared__ double sU[HM_W][296];    // group g at 147 g: per edge 3 rows of [H[6m..6m+5], bl[m]]
    __shared__ double part[HM_W][28];
    const int p = xcd_contiguous_item(ev.P);
    if (p >= ev.P)
        return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pj = min(lane / 9, 6), part9 = lane - 9 * (lane / 9);
    const bool writer = lane < 63;
    double* sT = reinterpret_cast<double*>(sT2[w]);
    double* su = sU[w];
    // operands of this lane in MFMA step q (K slot kq: column 4 q + kq of a group)
    const int m = lane & 15, kq = lane >> 4, h = m / 6, i = m - 6 * h;
    int offA[6], offB[6];
#pragma unroll
    for (int q = 0; q < 6; q++)
    {
        const int col = 4 * q + kq;
        offA[q] = (m < 12 && col < 21) ? 126 * h + 6 * col + i : HM_ZT;
        offB[q] = col >= 21 ? HM_ZU : m < 12 ? 147 * h + 7 * col + i : m < 14 ? 147 * (m - 12) + 7 * col + 6 : HM_ZU;
    }
    if (lane < 2)
        sT[HM_ZT + lane] = 0.0, su[HM_ZU + lane] = 0.0;
    const int i0 = ev.pose_ptr[p], i1 = ev.pose_ptr[p + 1];
    const int nch = (i1 - i0 + HM_CH - 1) / HM_CH;
    hsc_d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    int ea_next = 0, eb_next = 0;
    if (w < nch)
    {
        ea_next = ev.pose_edge[min(i0 + HM_CH * w + pj, i1 - 1)];
        eb_next = ev.pose_edge[min(i0 + HM_CH * w + 7 + pj, i1 - 1)];
    }
    for (int ch = w; ch < nch; ch += HM_W)
    {
        const int ia = i0 + HM_CH * ch + pj, ib = ia + 7;
        const int ea = ea_next, eb = eb_next;
        if (ch + HM_W < nch)
        {
            ea_next = ev.pose_edge[min(ia + HM_CH * HM_W, i1 - 1)];
            eb_next = ev.pose_edge[min(ib + HM_CH * HM_W, i1 - 1)];
        }
        const uint8_t fa = ev.flags[ea], fb = ev.flags[eb];
        const int la = ev.lm[ea], lb = ev.lm[eb];
        const bool acta = ia < i1 && !(fa & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE));
        const bool actb = ib < i1 && !(fb & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE));
        // fixed-landmark edges have l >= L: no bl entry (and they are never active)
        double2 ta = ld_pair(T, 9 * (size_t)ea + part9), tb = ld_pair(T, 9 * (size_t)eb + part9);
        const double2 ha = ld_pair(Hpl, 9 * (size_t)ea + part9), hb = ld_pair(Hpl, 9 * (size_t)eb + part9);
        // (an inactive edge contributes a zero T block, but 0 x whatever sits at bl[0] must stay 0: with no free
        // landmark at all bl is empty and that word is memory nobody wrote)
        // (loaded unconditionally — a conditional load is a branch on the dependent chain: 40 -> 46 us — and
        // selected afterwards)
        const double bla = bl[3 * (size_t)(acta ? la : 0) + min(part9, 2)];
        const double blb = bl[3 * (size_t)(actb ? lb : 0) + min(part9, 2)];
        const double ba = acta ? bla : 0.0, bb = actb ? blb : 0.0;
        if (!acta)
            ta = make_double2(0, 0); // the edge contributes nothing
        if (!actb)
            tb = make_double2(0, 0);
        if (writer)
        {
            sT2[w][9 * pj + part9] = ta, sT2[w][63 + 9 * pj + part9] = tb;
            const int k0 = 2 * part9, k1 = k0 + 1;
            const int u0 = 21 * pj + 7 * (k0 / 6) + k0 % 6, u1 = 21 * pj + 7 * (k1 / 6) + k1 % 6;
            su[u0] = ha.x, su[u1] = ha.y;
            su[147 + u0] = hb.x, su[147 + u1] = hb.y;
            if (part9 < 3)
                su[21 * pj + 7 * part9 + 6] = ba, su[147 + 21 * pj + 7 * part9 + 6] = bb;
        }
        wave_sync_lds0();
        double a[6], b[6];
#pragma unroll
        for (int q = 0; q < 6; q++)
            a[q] = sT[offA[q]], b[q] = su[offB[q]];
#pragma unroll
        for (int q = 0; q < 6; q += 2)
        {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q + 1], b[q + 1], acc1, 0, 0, 0);
        }
        wave_sync_lds0(); // the slot is rewritten by the next chunk
    }
    // D[row][col]: row = (lane >> 4) + 4 reg, col = lane & 15
#pragma unroll
    for (int q = 0; q < 4; q++)
        sT[16 * (kq + 4 * q) + m] = acc0[q] + acc1[q];
    wave_sync_lds0();
    if (lane < 27)
    {
        int r = 0, cc = 12; // lane < 21 -> (r, cc) of the upper triangle, else rhs row r
        if (lane < 21)
        {
            int k = lane;
            while (k >= 6 - r)
            {
                k -= 6 - r;
                r++;
            }
            cc = r + k;
        }
        else
            r = lane - 21;
        part[w][lane] = lane < 21 ? sT[16 * r + cc] + sT[16 * (6 + r) + 6 + cc] : sT[16 * r + 12] + sT[16 * (6 + r) + 13];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 36)
    {
        const int rr = t % 6, c = t / 6;
        const int a = rr < c ? rr : c, b = rr < c ? c : rr;
        const int k = tri6(a, b);
        double sum = 0;
        for (int q = 0; q < HM_W; q++)
            sum += part[q][k];
        double val = Hpp[36 * (size_t)p + t] - sum;
        if (rr == c)
            val += lambda_diag;
        Hsc[36 * (size_t)rowptr[p] + t] = val;
    }
    else if (t < 42)
    {
        const int k = 21 + (t - 36);
        double sum = 0;
        for (int q = 0; q < HM_W; q++)
            sum += part[q][k];
        bsc[6 * (size_t)p + (t - 36)] = bp[6 * (size_t)p + (t - 36)] - sum;
    }
}

// ---------------------------------------------------------------- Schur: off-diagonal --
// one wave per Hsc block k: Hsc[k] = - sum_{(ei,ej)} T[ei] Hpl[ej]^T
// (ref: computeHschureKernel .cu:1327-1345, which uses 36 atomics per product instead).
// The vector memory path handles roughly one cache line per ~4 cycles per CU, so what counts is
// the number of lines a load instruction touches.  A chunk of 7 products is fetched by ONE pair
// of 16-B loads: lanes 9j..9j+8 read the contiguous 144-B T block (resp. Hpl block) of product j
// — ~2 lines per product instead of 64 lines per instruction with a lane-per-product gather.
// The operands go through a per-wave LDS slot; then LANE = OUTPUT ELEMENT (r,c), 36 of 64 lanes,
// accumulates its own element over the chunk: no cross-lane reduction, summation in list order
// (bit-reproducible).  The next chunk's loads are in flight while the current one is consumed.
constexpr int OD_CH = 14; // products per chunk: two 7-product groups per load round
template <typename S>
__global__ __launch_bounds__(BS) void k_hsc_offdiag(int nblocks,
                                                    const int32_t* __restrict__ off_ptr,
                                                    const int32_t* __restrict__ off_ei,
                                                    const int32_t* __restrict__ off_ej,
                                                    const S* __restrict__ Hpl,
                                                    const S* __restrict__ T,
                                                    double* __restrict__ Hsc)
{
    __shared__ double2 stage[BS / 64][2][OD_CH * 9 + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // (blocks stay in dispatch order here: giving every XCD a contiguous range of Hsc blocks was
    // measured slower, 123 vs 112 us — the operands shared by neighbouring blocks are then no
    // longer fetched by several XCDs at about the same time, which is what makes the re-reads
    // Infinity-Cache hits)
    const int k = blockIdx.x * (BS / 64) + w;
    if (k >= nblocks)
        return; // whole wave
    const int lc = lane < 36 ? lane : 35; // idle lanes shadow lane 35
    const int r = lc % 6, c = lc / 6;
    const int pj = min(lane / 9, 6), part = lane - 9 * (lane / 9); // lane 63 shadows product 6
    const bool writer = lane < 63;
    const int beg = off_ptr[k], end = off_ptr[k + 1];
    const double* sT = reinterpret_cast<const double*>(stage[w][0]);
    const double* sH = reinterpret_cast<const double*>(stage[w][1]);
    double acc = 0;
    if (beg >= end)
    {
        if (lane < 36)
            Hsc[36 * (size_t)k + lane] = 0.0;
        return;
    }
    // software pipeline: index pairs two chunks ahead, operands one chunk ahead
    // (scalars, not arrays: the compiler parks small indexed arrays in scratch memory)
    int j0 = min(beg + pj, end - 1), j1 = min(beg + pj + 7, end - 1); // clamped: surplus lanes re-read the last product
    int ei0 = off_ei[j0], ej0 = off_ej[j0], ei1 = off_ei[j1], ej1 = off_ej[j1];
    double2 tv0 = ld_pair(T, 9 * (size_t)ei0 + part), hv0 = ld_pair(Hpl, 9 * (size_t)ej0 + part);
    double2 tv1 = ld_pair(T, 9 * (size_t)ei1 + part), hv1 = ld_pair(Hpl, 9 * (size_t)ej1 + part);
    j0 = min(beg + OD_CH + pj, end - 1), j1 = min(beg + OD_CH + pj + 7, end - 1);
    ei0 = off_ei[j0], ej0 = off_ej[j0], ei1 = off_ei[j1], ej1 = off_ej[j1];
    double2* w0 = &stage[w][0][9 * pj + part];
    double2* w1 = &stage[w][1][9 * pj + part];
    for (int idx = beg; idx < end; idx += OD_CH)
    {
        if (writer)
        {
            w0[0] = tv0, w1[0] = hv0;
            w0[63] = tv1, w1[63] = hv1;
        }
        if (idx + OD_CH < end)
        {
            tv0 = ld_pair(T, 9 * (size_t)ei0 + part), hv0 = ld_pair(Hpl, 9 * (size_t)ej0 + part);
            tv1 = ld_pair(T, 9 * (size_t)ei1 + part), hv1 = ld_pair(Hpl, 9 * (size_t)ej1 + part);
            j0 = min(idx + 2 * OD_CH + pj, end - 1), j1 = min(idx + 2 * OD_CH + pj + 7, end - 1);
            ei0 = off_ei[j0], ej0 = off_ej[j0], ei1 = off_ei[j1], ej1 = off_ej[j1];
        }
        wave_sync_lds();
        const int n = min(OD_CH, end - idx);
#pragma unroll
        for (int u = 0; u < OD_CH; u++)
            if (u < n) // wave-uniform
            {
                double s = sT[18 * u + r] * sH[18 * u + c];
                s = fma(sT[18 * u + 6 + r], sH[18 * u + 6 + c], s);
                s = fma(sT[18 * u + 12 + r], sH[18 * u + 12 + c], s);
                acc += s;
            }
        wave_sync_lds(); // the slot is rewritten by the next iteration
    }
    if (lane < 36)
        Hsc[36 * (size_t)k + lane] = -acc;
}

// ---------------------------------------------------------------- Schur: off-diagonal, matrix cores
// The sum over a block's products IS a small GEMM: Hsc[k] = -[T_1 T_2 ...] [H_1 H_2 ...]^T with the 6x3 blocks
// side by side, K = 3 x products.  k_hsc_offdiag forms it on the vector lanes from LDS: six 8-byte LDS reads
// per product and lane, and the 128 B/clk LDS port of the CU is what bounds it (14 products: 84 reads x 4 clk
// for every wave of the CU).  Here the staged chunk goes through v_mfma_f64_16x16x4: the 7 products of load
// group 0 in rows / columns 0..5 of the 16x16 tile, the 7 of group 1 in rows / columns 6..11 (a split of K in
// two: the result is D[0:6,0:6] + D[6:12,6:12]; the other entries of D are by-products nobody reads), 21
// columns of K per group = 6 MFMA steps with one LDS read per operand, lane and step — 12 reads per chunk
// instead of 84.  Fetch, staging and list order as in k_hsc_offdiag; the sum of a block runs in another
// (fixed) order, so the two kernels agree to rounding, not bit for bit.
template <typename S, bool XCD>
__global__ __launch_bounds__(BS) void k_hsc_offdiag_mfma(int nblocks,
                                                         const int32_t* __restrict__ off_ptr,
                                                         const int32_t* __restrict__ off_ei,
                                                         const int32_t* __restrict__ off_ej,
                                                         const S* __restrict__ Hpl,
                                                         const S* __restrict__ T,
                                                         double* __restrict__ Hsc)
{
    __shared__ double2 stage[BS / 64][2][OD_CH * 9 + 2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // XCD: every XCD (workgroups congruent mod 8) takes a contiguous range of blocks — rows of Hsc whose
    // blocks share their T operands then meet in one L2
    const int wg = XCD ? xcd_contiguous_item((int)gridDim.x) : (int)blockIdx.x;
    const int k = wg * (BS / 64) + w;
    if (k >= nblocks)
        return; // whole wave
    const int pj = min(lane / 9, 6), part = lane - 9 * (lane / 9); // lane 63 shadows product 6
    const bool writer = lane < 63;
    const int beg = off_ptr[k], end = off_ptr[k + 1];
    double* sT = reinterpret_cast<double*>(stage[w][0]);
    double* sH = reinterpret_cast<double*>(stage[w][1]);
    if (beg >= end)
    {
        if (lane < 36)
            Hsc[36 * (size_t)k + lane] = 0.0;
        return;
    }
    // operand of this lane in MFMA step q: row (A) / column (B) m of the tile, K slot kq: column 4q + kq of
    // group h = m / 6, element i = m % 6 of that column.  Slots past the 21 columns of a group and the tile
    // rows 12..15 read a zero kept behind the chunk.
    constexpr int ZERO = 2 * (OD_CH * 9); // doubles
    const int m = lane & 15, kq = lane >> 4, h = m / 6, i = m - 6 * h;
    int off[6];
#pragma unroll
    for (int q = 0; q < 6; q++)
        off[q] = (m < 12 && 4 * q + kq < 21) ? 126 * h + 6 * (4 * q + kq) + i : ZERO;
    if (lane < 2)
        sT[ZERO + lane] = 0.0, sH[ZERO + lane] = 0.0;
    // software pipeline: index pairs two chunks ahead, operands one chunk ahead
    int j0 = min(beg + pj, end - 1), j1 = min(beg + pj + 7, end - 1); // clamped: surplus lanes re-read the last product
    int ei0 = off_ei[j0], ej0 = off_ej[j0], ei1 = off_ei[j1], ej1 = off_ej[j1];
    double2 tv0 = ld_pair(T, 9 * (size_t)ei0 + part), hv0 = ld_pair(Hpl, 9 * (size_t)ej0 + part);
    double2 tv1 = ld_pair(T, 9 * (size_t)ei1 + part), hv1 = ld_pair(Hpl, 9 * (size_t)ej1 + part);
    j0 = min(beg + OD_CH + pj, end - 1), j1 = min(beg + OD_CH + pj + 7, end - 1);
    ei0 = off_ei[j0], ej0 = off_ej[j0], ei1 = off_ei[j1], ej1 = off_ej[j1];
    double2* w0 = &stage[w][0][9 * pj + part];
    double2* w1 = &stage[w][1][9 * pj + part];
    hsc_d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int idx = beg; idx < end; idx += OD_CH)
    {
        // a product past the end of the list contributes a zero T block (its clamped re-read is finite)
        if (idx + pj >= end)
            tv0 = make_double2(0, 0);
        if (idx + pj + 7 >= end)
            tv1 = make_double2(0, 0);
        if (writer)
        {
            w0[0] = tv0, w1[0] = hv0;
            w0[63] = tv1, w1[63] = hv1;
        }
        if (idx + OD_CH < end)
        {
            tv0 = ld_pair(T, 9 * (size_t)ei0 + part), hv0 = ld_pair(Hpl, 9 * (size_t)ej0 + part);
            tv1 = ld_pair(T, 9 * (size_t)ei1 + part), hv1 = ld_pair(Hpl, 9 * (size_t)ej1 + part);
            j0 = min(idx + 2 * OD_CH + pj, end - 1), j1 = min(idx + 2 * OD_CH + pj + 7, end - 1);
            ei0 = off_ei[j0], ej0 = off_ej[j0], ei1 = off_ei[j1], ej1 = off_ej[j1];
        }
        wave_sync_lds();
        double a[6], b[6];
#pragma unroll
        for (int q = 0; q < 6; q++)
            a[q] = sT[off[q]], b[q] = sH[off[q]];
#pragma unroll
        for (int q = 0; q < 6; q += 2)
        {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q + 1], b[q + 1], acc1, 0, 0, 0);
        }
        wave_sync_lds(); // the slot is rewritten by the next iteration
    }
    // D[row][col]: row = (lane >> 4) + 4 reg, col = lane & 15; out(r, c) = D[r][c] + D[6 + r][6 + c]
#pragma unroll
    for (int q = 0; q < 4; q++)
        sT[16 * (kq + 4 * q) + m] = acc0[q] + acc1[q];
    wave_sync_lds();
    if (lane < 36)
    {
        const int r = lane % 6, c = lane / 6;
        Hsc[36 * (size_t)k + lane] = -(sT[16 * r + c] + sT[16 * (6 + r) + 6 + c]);
    }
}

// ---------------------------------------------------------------- Schur: off-diagonal, row strips
// The same sums as k_hsc_offdiag — same products, same order, bit for bit — with ONE workgroup per block
// row p of Hsc instead of one wave per block anywhere.  Every product of the row has its T operand among
// the edges of pose p (T_e Hpl_e'^T: e an edge of p, e' the edge of e's landmark seen by the column's pose),
// so the T blocks of the pose's ~420 edges are staged in LDS ONCE per row (61 KB: each T block is read from
// memory once per Schur complement instead of once per product, 2.6x less on the kitti_00 shape) and only the
// Hpl operands are gathered per product — half the gather instructions and lines of the block-per-wave form,
// and the rows of a neighbourhood (same XCD: xcd_contiguous_item) gather the same Hpl blocks out of L2.
// The lists carry, beside the slot of the T edge, its position in the pose's edge list (off_pi, built once
// per structure: k_list_pos).  A row whose pose has more edges than fit (HS_CAP) gathers T as before.
constexpr int HS_BS = 1024;
constexpr int HS_W = HS_BS / 64;
constexpr int HS_CAP = 704; // T blocks of a pose that fit the LDS stage (99 KB)
constexpr size_t hs_lds_bytes() { return (size_t)HS_CAP * 18 * sizeof(double) + (size_t)HS_W * (OD_CH * 9 + 1) * sizeof(double2) + (size_t)HS_W * 16 * sizeof(int32_t); }

// pos[e] = position of slot e in the edge list of its pose
__global__ __launch_bounds__(BS) void k_pose_pos(EV ev, int n, int32_t* __restrict__ pos)
{
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n)
        return;
    const int e = ev.pose_edge[i];
    pos[e] = i - ev.pose_ptr[ev.pose[e]];
}
__global__ __launch_bounds__(BS) void k_list_pos(size_t M, const int32_t* __restrict__ off_ei,
                                                 const int32_t* __restrict__ pos, int32_t* __restrict__ off_pi)
{
    const size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    if (i < M)
        off_pi[i] = pos[off_ei[i]];
}

template <typename S>
__global__ __launch_bounds__(HS_BS) void k_hsc_offdiag_strip(EV ev, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ off_ptr,
                                                            const int32_t* __restrict__ off_ei,
                                                            const int32_t* __restrict__ off_pi,
                                                            const int32_t* __restrict__ off_ej,
                                                            const S* __restrict__ Hpl, const S* __restrict__ T,
                                                            double* __restrict__ Hsc)
{
    extern __shared__ double hs_lds[];
    double* sTrow = hs_lds;                                                        // [HS_CAP][18]
    double2* stage = reinterpret_cast<double2*>(hs_lds + (size_t)HS_CAP * 18);     // [HS_W][OD_CH*9+1]
    int32_t* spi = reinterpret_cast<int32_t*>(stage + HS_W * (OD_CH * 9 + 1));     // [HS_W][16]
    const int p = xcd_contiguous_item(ev.P);
    if (p >= ev.P)
        return;
    const int b0 = rowptr[p] + 1, b1 = rowptr[p + 1]; // the row's off-diagonal blocks
    if (b0 >= b1)
        return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i0 = ev.pose_ptr[p], np = ev.pose_ptr[p + 1] - i0;
    const bool in_lds = np <= HS_CAP; // (uniform)
    if (in_lds)
    { // the T blocks of the pose's edges, in list order: pair `q` of list entry `i` (9 pairs per 144-B block)
        double2* dst = reinterpret_cast<double2*>(sTrow);
        for (int idx = tid; idx < 9 * np; idx += HS_BS)
        {
            const int i = idx / 9, q = idx - 9 * i;
            dst[idx] = ld_pair(T, 9 * (size_t)ev.pose_edge[i0 + i] + q);
        }
    }
    __syncthreads();
    const int lc = lane < 36 ? lane : 35; // idle lanes shadow lane 35
    const int r = lc % 6, c = lc / 6;
    const int pj = min(lane / 9, 6), part = lane - 9 * (lane / 9); // lane 63 shadows product 6
    const bool writer = lane < 63;
    double2* st = stage + w * (OD_CH * 9 + 1);
    const double* sH = reinterpret_cast<const double*>(st);
    int32_t* pis = spi + w * 16;
    for (int k = b0 + w; k < b1; k += HS_W)
    {
        const int beg = off_ptr[k], end = off_ptr[k + 1];
        double acc = 0;
        // software pipeline as in k_hsc_offdiag: Hpl operands one chunk ahead, list entries two ahead
        int j0 = min(beg + pj, end - 1), j1 = min(beg + pj + 7, end - 1);
        int ej0 = off_ej[j0], ej1 = off_ej[j1];
        int pi0 = in_lds ? off_pi[j0] : off_ei[j0], pi1 = in_lds ? off_pi[j1] : off_ei[j1];
        double2 hv0 = ld_pair(Hpl, 9 * (size_t)ej0 + part), hv1 = ld_pair(Hpl, 9 * (size_t)ej1 + part);
        int pw0 = pi0, pw1 = pi1;
        j0 = min(beg + OD_CH + pj, end - 1), j1 = min(beg + OD_CH + pj + 7, end - 1);
        ej0 = off_ej[j0], ej1 = off_ej[j1];
        pi0 = in_lds ? off_pi[j0] : off_ei[j0], pi1 = in_lds ? off_pi[j1] : off_ei[j1];
        for (int idx = beg; idx < end; idx += OD_CH)
        {
            if (writer)
            {
                st[9 * pj + part] = hv0, st[63 + 9 * pj + part] = hv1;
                if (part == 0)
                    pis[pj] = pw0, pis[7 + pj] = pw1;
            }
            if (idx + OD_CH < end)
            {
                hv0 = ld_pair(Hpl, 9 * (size_t)ej0 + part), hv1 = ld_pair(Hpl, 9 * (size_t)ej1 + part);
                pw0 = pi0, pw1 = pi1;
                j0 = min(idx + 2 * OD_CH + pj, end - 1), j1 = min(idx + 2 * OD_CH + pj + 7, end - 1);
                ej0 = off_ej[j0], ej1 = off_ej[j1];
                pi0 = in_lds ? off_pi[j0] : off_ei[j0], pi1 = in_lds ? off_pi[j1] : off_ei[j1];
            }
            wave_sync_lds();
            const int n = min(OD_CH, end - idx);
#pragma unroll
            for (int u = 0; u < OD_CH; u++)
                if (u < n) // wave-uniform
                {
                    double t0, t1, t2;
                    if (in_lds)
                    {
                        const double* Tt = sTrow + 18 * pis[u];
                        t0 = Tt[r], t1 = Tt[6 + r], t2 = Tt[12 + r];
                    }
                    else
                    { // (a pose with more edges than the stage holds: its T blocks straight from memory)
                        const S* Tg = T + 18 * (size_t)pis[u];
                        t0 = (double)Tg[r], t1 = (double)Tg[6 + r], t2 = (double)Tg[12 + r];
                    }
                    double s = t0 * sH[18 * u + c];
                    s = fma(t1, sH[18 * u + 6 + c], s);
                    s = fma(t2, sH[18 * u + 12 + c], s);
                    acc += s;
                }
            wave_sync_lds(); // the slot is rewritten by the next iteration
        }
        if (lane < 36)
            Hsc[36 * (size_t)k + lane] = -acc;
    }
}

// ---------------------------------------------------------------- Schur: pose rows ------
// The whole block row p of the Schur complement in ONE workgroup (round 3; replaces k_schur_edges' T
// stream, k_hsc_offdiag and k_hsc_diag when the engine asks for it):
//     Hsc(p, p) = Hpp[p] (+ lambda I) - sum_e T_e Hpl_e^T        e = edges of pose p
//     Hsc(p, q) =                     - sum_e T_e Hpl_e'^T       e' = the edge of e's landmark seen by pose q > p
//     bsc[p]    = bp[p] - sum_e T_e bl[l(e)],                    T_e = Hpl_e (Hll + lambda I)^-1
// (ref: computeBschureKernel + computeHschureKernel, src/cuda/cuda_block_solver.cu:1286-1345.)
// Why: the destination-major gather (one wave per Hsc block over its product list) reads every T and Hpl
// block once per product — 2.6x the compulsory bytes on the kitti_00 shape, 6-8x in memory-side traffic —
// and runs at ~5 TB/s on that re-fetch traffic.  Here the row's accumulators live in LDS for the lifetime
// of the workgroup and the pose's edges are STREAMED through it once: Hpl_e is read once for the diagonal
// product and T_e is formed from it on the spot (T is never written or read); the partner blocks Hpl_e' are
// the slots right behind e in the landmark-major layout (contiguous), and the poses of a neighbourhood
// (same XCD: xcd_contiguous_item) re-read the same lines out of L2.  No product lists at all: the column of
// a product is looked up from the partner's pose index (a direct map for the near columns, binary search in
// the row's colind for the far ones).
// Mapping: a round = 28 edges of the pose (7 per wave, lanes 9j..9j+8 load edge j's 144-B block: ~2 cache
// lines per block), staged in LDS with their first HR_PM partner blocks; products are dealt to the four
// waves BY DESTINATION COLUMN (column j belongs to wave j mod 4: no two waves ever add to the same block,
// each block's products arrive in ascending landmark order: fixed order, bit-reproducible), lane = output
// element; the diagonal block and the rhs are summed in registers per wave and added in wave order.
constexpr int HR_BS = 256;
constexpr int HR_W = HR_BS / 64;
constexpr int HR_CH = 7;            // edges per wave and round
constexpr int HR_E = HR_W * HR_CH;  // edges per round
constexpr int HR_PM = 4;            // partner blocks of an edge staged per sub-round
constexpr int HR_NEAR = 1024;       // direct map q - p -> column position for the near columns
constexpr int HR_SH = 1 + HR_PM;    // blocks staged per edge: its own + partners
// doubles of dynamic LDS besides the nnz_cap x 36 accumulators
constexpr int hr_fixed_doubles() { return HR_E * 18 + HR_E * HR_SH * 18 + HR_E * 9 + HR_E * 3 + HR_W * 42 + HR_NEAR / 4 + HR_E * HR_PM / 2 + HR_E / 2 + 8; }

} // namespace
namespace cugo_k
{
bool schur_rows_usable(const cugo_hsc_struct& hs, int max_row_nnz);
}
namespace
{
using cugo_k::schur_rows_usable;
// per entry of the pose-major edge list: {slot, end of its landmark's slots, landmark, flags}
__global__ __launch_bounds__(BS) void k_pose_rec(EV ev, int n, int4* __restrict__ rec)
{
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n)
        return;
    const int e = ev.pose_edge[i];
    const int l = ev.lm[e];
    rec[i] = make_int4(e, ev.lm_ptr[l + 1], l, (int)ev.flags[e]);
}

// invHll = (Hll + lambda I)^-1 alone (the retry of a rejected trial: a new lambda, no new build pass)
__global__ __launch_bounds__(BS) void k_inv_hll(int L, double lambda, const double* __restrict__ Hll,
                                                double* __restrict__ invHll)
{
    const int l = blockIdx.x * BS + threadIdx.x;
    if (l >= L)
        return;
    const Sym3 iv = sym3_inv(Hll + 9 * (size_t)l, lambda);
    double* o = invHll + 9 * (size_t)l;
    o[0] = iv.b00, o[1] = iv.b01, o[2] = iv.b02;
    o[3] = iv.b01, o[4] = iv.b11, o[5] = iv.b12;
    o[6] = iv.b02, o[7] = iv.b12, o[8] = iv.b22;
}

template <typename S>
__global__ __launch_bounds__(HR_BS) void k_hsc_rows(EV ev, const int4* __restrict__ rec,
                                                    const int32_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ colind, double lambda_diag,
                                                    const double* __restrict__ Hpp, const double* __restrict__ bp,
                                                    const double* __restrict__ bl,
                                                    const double* __restrict__ invHll, const S* __restrict__ Hpl,
                                                    double* __restrict__ Hsc, double* __restrict__ bsc, int nnz_cap)
{
    extern __shared__ double hr_lds[];
    double* acc = hr_lds;                                  // [nnz_cap][36]; column 0 (the diagonal) unused
    double* sT = acc + 36 * (size_t)nnz_cap;               // [HR_E][18]
    double* sH = sT + HR_E * 18;                           // [HR_E][HR_SH][18]
    double* sInv = sH + HR_E * HR_SH * 18;                 // [HR_E][9]
    double* sBl = sInv + HR_E * 9;                         // [HR_E][3]
    double* dpart = sBl + HR_E * 3;                        // [HR_W][42]
    int16_t* near = reinterpret_cast<int16_t*>(dpart + HR_W * 42); // [HR_NEAR]
    int32_t* sdst = reinterpret_cast<int32_t*>(near + HR_NEAR);   // [HR_E][HR_PM] column of a partner product, -1: none
    int32_t* sact = sdst + HR_E * HR_PM;                           // [HR_E] edge contributes
    const int p = xcd_contiguous_item(ev.P);
    if (p >= ev.P)
        return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row0 = rowptr[p], nnz = rowptr[p + 1] - row0;
    for (int j = tid; j < nnz; j += HR_BS)
    {
        const int q = colind[row0 + j];
        if (q - p < HR_NEAR)
            near[q - p] = (int16_t)j;
    }
    for (int i = tid; i < 36 * nnz; i += HR_BS)
        acc[i] = 0.0;
    const int i0 = ev.pose_ptr[p], i1 = ev.pose_ptr[p + 1];
    // staging role: lane = 9 pj + part loads the pair `part` of edge pj of this wave's seven
    const int pj = min(lane / 9, HR_CH - 1), part = lane - 9 * (lane / 9);
    const bool writer = lane < 9 * HR_CH;
    const int u = HR_CH * w + pj; // edge of the round this lane stages
    // product role: lane = output element (r, c) of a 6x6 product; the diagonal block is formed as its
    // upper triangle mirrored (exactly symmetric); lanes 36..41: the rhs
    const int lc = lane < 36 ? lane : 35;
    const int r6 = lc % 6, c6 = lc / 6;
    const int dr = min(r6, c6), dc = max(r6, c6);
    const int rb = lane >= 36 && lane < 42 ? lane - 36 : 0;
    double dacc = 0.0, bacc = 0.0;
    constexpr uint8_t NOT_FF = CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE;
    __syncthreads();
    const int nrounds = (i1 - i0 + HR_E - 1) / HR_E;
    int4 rc = make_int4(0, 0, 0, NOT_FF);
    if (nrounds > 0)
        rc = rec[min(i0 + u, i1 - 1)];
    for (int rd = 0; rd < nrounds; rd++)
    {
        const int idx = i0 + HR_E * rd + u;
        const int e = rc.x, lend = rc.y;
        const bool act = writer && idx < i1 && !(rc.w & NOT_FF);
        const int l = act ? rc.z : 0;
        const int np = act ? lend - e - 1 : 0; // slots behind e in its landmark (inactive ones included)
        if (rd + 1 < nrounds) // the next round's records: in flight during this round
            rc = rec[min(idx + HR_E, i1 - 1)];
        // ---- loads of this round: own block, invHll and bl of its landmark, the first HR_PM partners
        const double2 hv = ld_pair(Hpl, 9 * (size_t)e + part);
        const double ivv = invHll[9 * (size_t)l + part];
        const double blv = bl[3 * (size_t)l + min(part, 2)];
        for (int t0 = 0;; t0 += HR_PM)
        {
            double2 pv[HR_PM];
#pragma unroll
            for (int t = 0; t < HR_PM; t++)
            {
                const int es = e + 1 + t0 + t;
                pv[t] = ld_pair(Hpl, 9 * (size_t)(es < lend ? es : e) + part);
            }
            int dstj = -1;
            if (part < HR_PM)
            { // lane `part` of the group looks up the column of partner t0 + part
                const int es = e + 1 + t0 + part;
                if (act && es < lend && !(ev.flags[es] & NOT_FF))
                {
                    const int q = ev.pose[es];
                    if (q - p < HR_NEAR)
                        dstj = near[q - p];
                    else
                    { // far column (loop closure): binary search in the row's sorted column list
                        int lo = 0, hi = nnz - 1;
                        while (lo < hi)
                        {
                            const int mid = (lo + hi) >> 1;
                            if (colind[row0 + mid] < q)
                                lo = mid + 1;
                            else
                                hi = mid;
                        }
                        dstj = lo;
                    }
                }
            }
            // ---- stage
            if (writer)
            {
                double* h0 = sH + (u * HR_SH) * 18 + 2 * part;
                if (t0 == 0)
                {
                    h0[0] = hv.x, h0[1] = hv.y;
                    sInv[u * 9 + part] = ivv;
                    if (part < 3)
                        sBl[u * 3 + part] = blv;
                    if (part == 0)
                        sact[u] = act ? 1 : 0;
                }
#pragma unroll
                for (int t = 0; t < HR_PM; t++)
                    h0[18 * (1 + t)] = pv[t].x, h0[18 * (1 + t) + 1] = pv[t].y;
                if (part < HR_PM)
                    sdst[u * HR_PM + part] = dstj;
            }
            if (t0 == 0)
            {
                wave_sync_lds();
                if (writer)
                { // T = Hpl invHll: entries 2 part, 2 part + 1 (the operation order of k_schur_edges)
                    const double* H = sH + (u * HR_SH) * 18;
                    const double* iv = sInv + u * 9;
#pragma unroll
                    for (int k = 0; k < 2; k++)
                    {
                        const int i = 2 * part + k, r = i % 6, c = i / 6;
                        sT[u * 18 + i] = H[r] * iv[c] + H[6 + r] * iv[3 + c] + H[12 + r] * iv[6 + c];
                    }
                }
            }
            __syncthreads();
            // ---- products
            if (t0 == 0)
            { // the diagonal block and the rhs: the seven edges of this wave, in edge order
#pragma unroll
                for (int k = 0; k < HR_CH; k++)
                {
                    const int uu = HR_CH * w + k;
                    if (sact[uu]) // wave-uniform
                    {
                        const double* T = sT + uu * 18;
                        const double* H = sH + (uu * HR_SH) * 18;
                        double sd = T[dr] * H[dc];
                        sd = fma(T[6 + dr], H[6 + dc], sd);
                        sd = fma(T[12 + dr], H[12 + dc], sd);
                        dacc += sd;
                        const double* b3 = sBl + uu * 3;
                        double sb = T[rb] * b3[0];
                        sb = fma(T[6 + rb], b3[1], sb);
                        sb = fma(T[12 + rb], b3[2], sb);
                        bacc += sb;
                    }
                }
            }
            { // partner products whose column belongs to this wave (column j -> wave j mod 4)
                const int d0 = sdst[lane], d1 = lane < HR_E * HR_PM - 64 ? sdst[64 + lane] : -1;
                unsigned long long m0 = __ballot(d0 >= 0 && (d0 & (HR_W - 1)) == w);
                unsigned long long m1 = __ballot(d1 >= 0 && (d1 & (HR_W - 1)) == w);
                while (m0 | m1)
                {
                    int i;
                    if (m0)
                    {
                        i = __builtin_ctzll(m0);
                        m0 &= m0 - 1;
                    }
                    else
                    {
                        i = 64 + __builtin_ctzll(m1);
                        m1 &= m1 - 1;
                    }
                    const int uu = i / HR_PM, t = i - HR_PM * uu;
                    const int j = sdst[i];
                    const double* T = sT + uu * 18;
                    const double* H = sH + (uu * HR_SH + 1 + t) * 18;
                    double sp = T[r6] * H[c6];
                    sp = fma(T[6 + r6], H[6 + c6], sp);
                    sp = fma(T[12 + r6], H[12 + c6], sp);
                    if (lane < 36)
                        acc[36 * j + lane] += sp;
                }
            }
            // more partners than staged so far in any edge of the round?
            if (!__syncthreads_or(np > t0 + HR_PM))
                break;
        }
    }
    // ---- finish: diagonal block and rhs in wave order, off-diagonal blocks straight from the accumulators
    if (lane < 36)
        dpart[w * 42 + lane] = dacc;
    else if (lane < 42)
        dpart[w * 42 + lane] = bacc;
    __syncthreads();
    if (tid < 36)
    {
        double sum = 0;
        for (int q = 0; q < HR_W; q++)
            sum += dpart[q * 42 + tid];
        double val = Hpp[36 * (size_t)p + tid] - sum;
        if (tid % 7 == 0)
            val += lambda_diag;
        Hsc[36 * (size_t)row0 + tid] = val;
    }
    else if (tid < 42)
    {
        double sum = 0;
        for (int q = 0; q < HR_W; q++)
            sum += dpart[q * 42 + tid];
        bsc[6 * (size_t)p + (tid - 36)] = bp[6 * (size_t)p + (tid - 36)] - sum;
    }
    for (int i = 36 + tid; i < 36 * nnz; i += HR_BS)
        Hsc[36 * (size_t)row0 + i] = -acc[i];
}

// ---------------------------------------------------------------- back-substitution ----
// xl = invHll (bl - sum Hpl^T xp), landmark update, scale partials
// (ref: schurComplementPostKernel .cu:1419, updateLandmarksKernel .cu:1457,
//  computeScaleKernel .cu:1471 — fused)
// One workgroup per 256 consecutive edges.  The 256 Hpl blocks (36 KB, contiguous) are
// streamed into LDS with fully coalesced 16-B loads — a lane-per-landmark walk over its edges
// touches 64 cache lines per load instruction and is bound by the vector memory path, not by
// HBM.  Lane e then forms Hpl[e]^T xp[pose(e)] from LDS, and the lane holding a landmark's
// first edge subtracts the contributions of its edges in edge order (fixed order) and
// finishes the landmark.  Edges of a landmark beyond the block (never with the engine's
// padded layout) are recomputed from global memory by the owner.
template <typename S>
__device__ __forceinline__ void hplT_x(const S* __restrict__ H, const double* __restrict__ x,
                                       double& s0, double& s1, double& s2)
{
    s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int m = 0; m < 6; m++)
    {
        const double xm = x[m];
        s0 += (double)H[m] * xm;
        s1 += (double)H[6 + m] * xm;
        s2 += (double)H[12 + m] * xm;
    }
}

// pose update + scale partials of one 256-pose block (ref: updatePosesKernel .cu:1444)
__device__ __forceinline__ void dev_update_poses(int blk, int nP, double lambda,
                                                 const double* __restrict__ xp,
                                                 const double* __restrict__ bp,
                                                 const double* __restrict__ poses_in,
                                                 double* __restrict__ poses_out,
                                                 double* __restrict__ partials, double* sm)
{
    const int p = blk * BS + threadIdx.x;
    double sc = 0;
    if (p < nP)
    {
        double dx[6];
#pragma unroll
        for (int i = 0; i < 6; i++)
        {
            dx[i] = xp[6 * (size_t)p + i];
            sc += dx[i] * (lambda * dx[i] + bp[6 * (size_t)p + i]);
        }
        pose_exp_update(dx, poses_in + 7 * (size_t)p, poses_out + 7 * (size_t)p);
    }
    sc = block_sum(sc, sm);
    if (threadIdx.x == 0)
        partials[blk] = sc;
}

template <typename S>
__global__ __launch_bounds__(BS) void k_backsubst_landmarks(
    EV ev, double lambda, const double* __restrict__ invHll, const double* __restrict__ bl,
    const S* __restrict__ Hpl, const double* __restrict__ xp, double* __restrict__ xl,
    const double* __restrict__ lms_in, double* __restrict__ lms_out,
    double* __restrict__ partials, int nbl, double lambda_pose, const double* __restrict__ bp,
    const double* __restrict__ poses_in, double* __restrict__ poses_out, const double* __restrict__ lmrec)
{
    // lmrec != nullptr: the one-stream form of the fused iteration (k_build_edges): `Hpl` holds G = Hpl L^-T and the
    // landmark's line {L^-1, y = L^-1 bl}:  dx_l = L^-T (y - sum G_e^T dx_p)  in place of  invHll (bl - sum Hpl_e^T dx_p)
    __shared__ double sm[BS / 64];
    __shared__ double2 hs[BS * 9 + 1];
    // the per-slot products Hpl^T x take the place of the staged blocks once every lane has consumed its own
    // (37 KB of LDS per workgroup instead of 43: four workgroups per CU instead of three)
    double(*cs)[BS] = reinterpret_cast<double(*)[BS]>(hs);
    if ((int)blockIdx.x >= nbl)
    { // pose update + its scale partials ride in the same launch (ref: updatePosesKernel .cu:1444)
        dev_update_poses(blockIdx.x - nbl, ev.P, lambda_pose, xp, bp, poses_in, poses_out,
                         partials + nbl, sm);
        return;
    }
    const int t = threadIdx.x;
    const int ebase = blockIdx.x * BS;
    const int e = ebase + t;
    { // stream the block's Hpl slots into LDS (zeros past the end)
        const long nvalid = 9L * max(0, min(BS, ev.E - ebase));
        // a workgroup past the last edge slot (the launch also covers the landmarks without edges; a SHARD
        // can hold fewer slots than the graph has landmarks) reads slot 0: in bounds, never used
        const size_t pbase = nvalid > 0 ? 9 * (size_t)ebase : 0;
        double2 v[9];
#pragma unroll
        for (int i = 0; i < 9; i++)
        {
            const int idx = i * BS + t;
            v[i] = ld_pair(Hpl, pbase + (size_t)min((long)idx, max(nvalid - 1, 0L)));
        }
#pragma unroll
        for (int i = 0; i < 9; i++)
            hs[i * BS + t] = v[i];
    }
    int l = -1;
    bool act = false;
    double x[6] = {0, 0, 0, 0, 0, 0};
    if (e < ev.E)
    {
        l = ev.lm[e];
        const uint8_t fl = ev.flags[e];
        act = !(fl & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE));
        if (act)
        {
            const double2* px = reinterpret_cast<const double2*>(xp + 6 * (size_t)ev.pose[e]);
            const double2 a = px[0], b = px[1], c = px[2];
            x[0] = a.x, x[1] = a.y, x[2] = b.x, x[3] = b.y, x[4] = c.x, x[5] = c.y;
        }
    }
    __syncthreads();
    {
        double s0 = 0, s1 = 0, s2 = 0;
        if (act)
            hplT_x(reinterpret_cast<const double*>(hs) + 18 * t, x, s0, s1, s2);
        __syncthreads(); // every lane has read its block: the area becomes cs
        cs[0][t] = s0, cs[1][t] = s1, cs[2][t] = s2;
    }
    __syncthreads();
    double sc = 0;
    if (l >= 0 && l < ev.L && ev.lm_ptr[l] == e)
    { // owner of landmark l
        const double b0 = bl[3 * (size_t)l], b1 = bl[3 * (size_t)l + 1], b2 = bl[3 * (size_t)l + 2];
        const double* lr = lmrec ? lmrec + 16 * (size_t)l : nullptr;
        double c0 = lr ? lr[6] : b0, c1 = lr ? lr[7] : b1, c2 = lr ? lr[8] : b2;
        const int e1 = ev.lm_ptr[l + 1];
        const int bend = min(e1, ebase + BS);
        for (int ee = e; ee < bend; ee++)
            c0 -= cs[0][ee - ebase], c1 -= cs[1][ee - ebase], c2 -= cs[2][ee - ebase];
        for (int ee = bend; ee < e1; ee++)
        { // beyond the block: straight from global memory
            const uint8_t fl = ev.flags[ee];
            if (fl & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE))
                continue;
            double s0, s1, s2;
            hplT_x(Hpl + 18 * (size_t)ee, xp + 6 * (size_t)ev.pose[ee], s0, s1, s2);
            c0 -= s0, c1 -= s1, c2 -= s2;
        }
        double x0, x1, x2;
        if (lr)
        { // L^-T c;  L^-1: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2)
            x0 = lr[0] * c0 + lr[1] * c1 + lr[3] * c2;
            x1 = lr[2] * c1 + lr[4] * c2;
            x2 = lr[5] * c2;
        }
        else
        {
            const double* iv = invHll + 9 * (size_t)l;
            x0 = iv[0] * c0 + iv[3] * c1 + iv[6] * c2;
            x1 = iv[1] * c0 + iv[4] * c1 + iv[7] * c2;
            x2 = iv[2] * c0 + iv[5] * c1 + iv[8] * c2;
        }
        xl[3 * (size_t)l] = x0, xl[3 * (size_t)l + 1] = x1, xl[3 * (size_t)l + 2] = x2;
        lms_out[3 * (size_t)l] = lms_in[3 * (size_t)l] + x0;
        lms_out[3 * (size_t)l + 1] = lms_in[3 * (size_t)l + 1] + x1;
        lms_out[3 * (size_t)l + 2] = lms_in[3 * (size_t)l + 2] + x2;
        sc = x0 * (lambda * x0 + b0) + x1 * (lambda * x1 + b1) + x2 * (lambda * x2 + b2);
    }
    if (e < ev.L && ev.lm_ptr[e] == ev.lm_ptr[e + 1])
    { // landmark without any edge: zero step
        xl[3 * (size_t)e] = 0, xl[3 * (size_t)e + 1] = 0, xl[3 * (size_t)e + 2] = 0;
        lms_out[3 * (size_t)e] = lms_in[3 * (size_t)e];
        lms_out[3 * (size_t)e + 1] = lms_in[3 * (size_t)e + 1];
        lms_out[3 * (size_t)e + 2] = lms_in[3 * (size_t)e + 2];
    }
    sc = block_sum(sc, sm);
    if (threadIdx.x == 0)
        partials[blockIdx.x] = sc;
}

inline int div_up(long a, int b) { return (int)((a + b - 1) / b); }

} // namespace

namespace cugo_k
{

static LaunchHook* g_hook = nullptr;
void set_launch_hook(LaunchHook* h) { g_hook = h; }
LaunchHook* launch_hook() { return g_hook; }

// scratch = [block partials | per-edge records of the build pass (8 doubles per slot)]
// block partials: the scale partials of the update pass (one per 256 edge slots / landmarks + one per 256 poses) at the
// start and, behind them (spec_chi_offset), the chi2 partials of a build pass that ends a trial (launch_build:
// chi_behind_scale)
static size_t spec_chi_offset(int n_edges, int n_poses, int n_landmarks)
{
    return (size_t)div_up(n_edges > n_landmarks ? n_edges : n_landmarks, BS) + div_up(n_poses, BS) + 64;
}
static size_t scratch_partials(int n_edges, int n_poses, int n_landmarks)
{
    const size_t n = 2 * (size_t)div_up(n_edges > n_landmarks ? n_edges : n_landmarks, BS) + div_up(n_poses, BS) + 4096;
    return (n + 31) & ~size_t(31); // the records start 256-byte aligned
}
size_t reduce_scratch_doubles(int n_edges, int n_poses, int n_landmarks)
{
    return scratch_partials(n_edges, n_poses, n_landmarks) + 8 * ((size_t)n_edges + BS);
}

void launch_errors(hipStream_t s, const cugo_edges& e, const double* d_poses, const double* d_lms,
                   cugo_robust rk, ReduceScratch rs, double* d_chi)
{
    const EV ev = make_ev(e);
    const int nb = div_up(ev.E, BS);
    if (nb > 0)
        CUGO_LAUNCH(k_errors, dim3(nb), dim3(BS), 0, s, ev, d_poses, d_lms,
                           Robust2{{rk.type, rk.delta}, {rk.type_stereo, rk.delta_stereo}}, rs.d_partials);
    CUGO_LAUNCH(k_sum_partials, dim3(1), dim3(SP_BS), 0, s, rs.d_partials, nb, d_chi);
}

void launch_errors_tail(hipStream_t s, const cugo_edges& e, const double* d_poses, const double* d_lms,
                        cugo_robust rk, ReduceScratch rs, int n_scale_partials, double* d_out,
                        const double* d_flag, double* h_out, double seq, unsigned* d_done)
{
    const EV ev = make_ev(e);
    const int nb = div_up(ev.E, BS);
    // the scale partials of the update pass sit at the start of the scratch: the error pass uses the
    // record area behind the partial slots (free between two build passes)
    double* d_chi_part = rs.d_partials + scratch_partials(ev.E, ev.P, ev.L);
    if (nb > 0)
        CUGO_LAUNCH(k_errors, dim3(nb), dim3(BS), 0, s, ev, d_poses, d_lms,
                           Robust2{{rk.type, rk.delta}, {rk.type_stereo, rk.delta_stereo}}, d_chi_part);
    CUGO_LAUNCH(k_sum_partials2, dim3(2), dim3(SP_BS), 0, s, d_chi_part, nb, rs.d_partials, n_scale_partials,
                d_out, d_flag, h_out, seq, d_done);
}

void launch_edge_chi(hipStream_t s, const cugo_edges& e, const double* d_poses, const double* d_lms,
                     cugo_robust rk, double* d_chi_e)
{
    const EV ev = make_ev(e);
    const int nb = div_up(ev.E, BS);
    if (nb > 0)
        CUGO_LAUNCH(k_edge_chi, dim3(nb), dim3(BS), 0, s, ev, d_poses, d_lms,
                           Robust2{{rk.type, rk.delta}, {rk.type_stereo, rk.delta_stereo}}, d_chi_e);
}

// kernel templated on the block storage type S: the launch hook sees the plain kernel name
#define CUGO_LAUNCH_T(kernel, S, grid, block, lds, stream, ...)                     \
    do                                                                              \
    {                                                                               \
        ::cugo_k::LaunchScope _scope(#kernel, stream);                              \
        hipLaunchKernelGGL(kernel<S>, grid, block, lds, stream, __VA_ARGS__);       \
    } while (0)

template <typename S>
static void launch_build_t(hipStream_t s, const cugo_edges& e, const double* d_poses, const double* d_lms,
                           cugo_robust rk, double* d_Hpp, double* d_bp, double* d_Hll, double* d_bl,
                           S* d_Hpl, ReduceScratch rs, double* d_chi, double fuse_lambda, double* d_invHll,
                           S* d_T, double* d_lmrec, bool skip_poses, bool chi_behind_scale)
{
    const EV ev = make_ev(e);
    const Robust2 r{{rk.type, rk.delta}, {rk.type_stereo, rk.delta_stereo}};
    const int nb = div_up(ev.E > ev.L ? ev.E : ev.L, BS); // also covers the edgeless landmarks
    double* d_rec = rs.d_partials + scratch_partials(ev.E, ev.P, ev.L);
    // chi_behind_scale: this build pass ends a trial (launch_trial_tail_from_build sums its chi2 partials together with
    // the scale partials of the update pass, which sit at the start of the scratch)
    double* d_chi_part = rs.d_partials + (chi_behind_scale ? spec_chi_offset(ev.E, ev.P, ev.L) : 0);
    if (nb > 0)
        CUGO_LAUNCH_T(k_build_edges, S, dim3(nb), dim3(BS), 0, s, ev, d_poses, d_lms, r, d_Hpl, d_Hll,
                      d_bl, d_rec, d_chi_part, d_invHll ? fuse_lambda : -1.0, d_invHll, d_T,
                      d_invHll && d_T ? d_lmrec : nullptr);
    if (d_chi)
        CUGO_LAUNCH(k_sum_partials, dim3(1), dim3(SP_BS), 0, s, d_chi_part, nb, d_chi);
    // skip_poses: the Schur complement of this lambda forms Hsc's diagonal blocks, bp and bsc from the records itself
    // (launch_schur with SchurRows::d_lmrec); Hpp is then not written
    if (ev.P > 0 && !(skip_poses && d_invHll && d_T && d_lmrec))
        CUGO_LAUNCH(k_build_poses, dim3(ev.P), dim3(BS), 0, s, ev, d_rec, d_Hpp, d_bp);
}

void launch_build(hipStream_t s, const cugo_edges& e, const double* d_poses, const double* d_lms,
                  cugo_robust rk, double* d_Hpp, double* d_bp, double* d_Hll, double* d_bl,
                  void* d_Hpl, ReduceScratch rs, double* d_chi, double fuse_lambda, double* d_invHll, void* d_T,
                  double* d_lmrec, bool skip_poses, bool chi_behind_scale)
{
    if (e.block_f32)
        launch_build_t(s, e, d_poses, d_lms, rk, d_Hpp, d_bp, d_Hll, d_bl, static_cast<float*>(d_Hpl), rs, d_chi,
                       fuse_lambda, d_invHll, static_cast<float*>(d_T), d_lmrec, skip_poses, chi_behind_scale);
    else
        launch_build_t(s, e, d_poses, d_lms, rk, d_Hpp, d_bp, d_Hll, d_bl, static_cast<double*>(d_Hpl), rs, d_chi,
                       fuse_lambda, d_invHll, static_cast<double*>(d_T), d_lmrec, skip_poses, chi_behind_scale);
}

// The end of a trial whose chi2 comes out of the NEXT iteration's build pass (launch_build with chi_behind_scale, queued
// at the trial's estimates before its result is known) instead of an error pass of its own: the same two-workgroup
// reduction as launch_errors_tail — chi2 partials + scale partials -> F-hat, scale, flag, sequence number in pinned memory.
// The partials are those k_errors would have written (same residual code, same 256-slot blocks, zeros behind them).
void launch_trial_tail_from_build(hipStream_t s, const cugo_edges& e, ReduceScratch rs, int n_scale_partials,
                                  double* d_out, const double* d_flag, double* h_out, double seq, unsigned* d_done)
{
    const EV ev = make_ev(e);
    const int nb = div_up(ev.E > ev.L ? ev.E : ev.L, BS);
    CUGO_LAUNCH(k_sum_partials2, dim3(2), dim3(SP_BS), 0, s, rs.d_partials + spec_chi_offset(ev.E, ev.P, ev.L), nb,
                rs.d_partials, n_scale_partials, d_out, d_flag, h_out, seq, d_done);
}

void launch_max_diagonal(hipStream_t s, const double* d_Hpp, int nP, const double* d_Hll, int nL,
                         ReduceScratch rs, double* d_out)
{
    const long n = 6L * nP + 3L * nL;
    int nb = div_up(n, BS);
    if (nb > 1024)
        nb = 1024;
    if (nb < 1)
        nb = 1;
    CUGO_LAUNCH(k_max_diag, dim3(nb), dim3(BS), 0, s, d_Hpp, nP, d_Hll, nL, rs.d_partials);
    CUGO_LAUNCH(k_max_partials, dim3(1), dim3(BS), 0, s, rs.d_partials, nb, d_out);
}

template <typename S>
static void launch_schur_t(hipStream_t s, const cugo_edges& e, const cugo_hsc_struct& hs, double lambda,
                           int damp_hsc_diag, const double* d_Hpp, const double* d_bp, const double* d_Hll,
                           const double* d_bl, const S* d_Hpl, double* d_invHll, S* d_T,
                           double* d_bsc, double* d_Hsc, bool have_T, SchurRows rows)
{
    const EV ev = make_ev(e);
    if (hs.d_grp_ptr && hs.n_groups == div_up(ev.E, BS))
    {
        const SchurPlanDev pl{hs.d_grp_ptr, hs.d_grp_nwave, hs.d_slot_rhs, hs.d_slot_ptr, hs.d_prod, hs.d_part_H, hs.d_part_b};
        if (ev.E > 0)
            CUGO_LAUNCH_T(k_schur_fused, S, dim3(hs.n_groups), dim3(SF_BS), 0, s, ev, lambda, d_Hll, d_bl, d_Hpl,
                          d_invHll, d_T, pl);
        if (hs.n_blocks > 0)
            CUGO_LAUNCH(k_hsc_reduce, dim3(div_up(hs.n_blocks, BS / 64)), dim3(BS), 0, s, hs.n_blocks, hs.d_red_ptr,
                        hs.d_red_slot, hs.d_slot_rhs, hs.d_blk_pose, hs.d_part_H, hs.d_part_b,
                        damp_hsc_diag ? lambda : 0.0, d_Hpp, d_bp, d_Hsc, d_bsc);
        return;
    }
    if (rows.d_pose_rec && schur_rows_usable(hs, rows.max_row_nnz))
    {
        const size_t lds = (36 * (size_t)std::max(rows.max_row_nnz, 1) + hr_fixed_doubles()) * sizeof(double);
        { // the whole block row of a pose in one workgroup: no T stream, no product lists (k_hsc_rows)
            if (ev.L > 0 && !have_T)
                CUGO_LAUNCH(k_inv_hll, dim3(div_up(ev.L, BS)), dim3(BS), 0, s, ev.L, lambda, d_Hll, d_invHll);
            if (ev.P > 0)
            {
                static bool attr_set[2] = {false, false};
                if (!attr_set[sizeof(S) == 4])
                {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hsc_rows<S>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                    attr_set[sizeof(S) == 4] = true;
                }
                CUGO_LAUNCH_T(k_hsc_rows, S, dim3(xcd_grid(ev.P)), dim3(HR_BS), lds, s, ev,
                              reinterpret_cast<const int4*>(rows.d_pose_rec), hs.d_rowptr, hs.d_colind,
                              damp_hsc_diag ? lambda : 0.0, d_Hpp, d_bp, d_bl, (const double*)d_invHll, d_Hpl, d_Hsc,
                              d_bsc, std::max(rows.max_row_nnz, 1));
            }
            return;
        }
    }
    if (ev.E > 0 && !have_T) // have_T: the build pass left invHll and T for this lambda (launch_build)
        CUGO_LAUNCH_T(k_schur_edges, S, dim3(div_up(ev.E, BS)), dim3(BS), 0, s, ev, lambda, d_Hll,
                      d_Hpl, d_invHll, d_T);
    if (hs.n_blocks > 0 && rows.d_off_pi && hs.d_rowptr && ev.P > 0)
    { // one workgroup per block row, the row's T blocks staged in LDS (k_hsc_offdiag_strip)
        static bool attr_set[2] = {false, false};
        if (!attr_set[sizeof(S) == 4])
        {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hsc_offdiag_strip<S>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)hs_lds_bytes());
            attr_set[sizeof(S) == 4] = true;
        }
        CUGO_LAUNCH_T(k_hsc_offdiag_strip, S, dim3(xcd_grid(ev.P)), dim3(HS_BS), hs_lds_bytes(), s, ev, hs.d_rowptr,
                      hs.d_off_ptr, hs.d_off_ei, rows.d_off_pi, hs.d_off_ej, d_Hpl, (const S*)d_T, d_Hsc);
    }
    else if (hs.n_blocks > 0 && rows.mfma)
    { // (rows.mfma == 0, CUGO_HSC_MFMA=0: the vector-lane kernels below; 2: only the off-diagonal kernel here)
        // a contiguous range of blocks per XCD (CUGO_HSC_XCD=0: dispatch order).  In-process A/B: 39.35 vs 41.80 ms
        // per step on the 10k-pose graph, 11.38 vs 11.36 ms on the kitti_00 shape — with the LDS port out of the
        // way the kernel is bound by its L2 misses, and rows that share T operands now meet in one L2 (the
        // vector-lane kernel, bound by LDS reads, was slower with this mapping: 123 vs 112 us)
        ::cugo_k::LaunchScope _scope("k_hsc_offdiag_mfma", s);
        if (have_T && rows.d_lmrec) // one-stream form (k_build_edges with lmrec): T holds G, Hsc_ij = - sum G_i G_j^T
            d_Hpl = (const S*)d_T;
        if (rows.xcd)
            hipLaunchKernelGGL((k_hsc_offdiag_mfma<S, true>), dim3(xcd_grid(div_up(hs.n_blocks, BS / 64))), dim3(BS), 0, s,
                               hs.n_blocks, hs.d_off_ptr, hs.d_off_ei, hs.d_off_ej, d_Hpl, (const S*)d_T, d_Hsc);
        else
            hipLaunchKernelGGL((k_hsc_offdiag_mfma<S, false>), dim3(div_up(hs.n_blocks, BS / 64)), dim3(BS), 0, s,
                               hs.n_blocks, hs.d_off_ptr, hs.d_off_ei, hs.d_off_ej, d_Hpl, (const S*)d_T, d_Hsc);
    }
    else if (hs.n_blocks > 0)
        CUGO_LAUNCH_T(k_hsc_offdiag, S, dim3(div_up(hs.n_blocks, BS / 64)), dim3(BS), 0, s,
                      hs.n_blocks, hs.d_off_ptr, hs.d_off_ei, hs.d_off_ej, d_Hpl, (const S*)d_T, d_Hsc);
    if (ev.P > 0 && have_T && rows.d_lmrec)
    { // fused iteration: diagonal blocks, bp and bsc from the build pass's records (its pose pass was skipped)
        const double* d_rec = rows.rs.d_partials + scratch_partials(ev.E, ev.P, ev.L);
        CUGO_LAUNCH(k_pose_schur, dim3(xcd_grid(ev.P)), dim3(PS_BS), 0, s, ev, d_rec, rows.d_lmrec, rows.d_poses,
                      hs.d_rowptr, damp_hsc_diag ? lambda : 0.0, d_Hsc, rows.d_bp_out, d_bsc);
    }
    else if (ev.P > 0 && rows.mfma == 1)
        CUGO_LAUNCH_T(k_hsc_diag_mfma, S, dim3(xcd_grid(ev.P)), dim3(HM_BS), 0, s, ev,
                      hs.d_rowptr, damp_hsc_diag ? lambda : 0.0, d_Hpp, d_bp, d_bl, d_Hpl, (const S*)d_T,
                      d_Hsc, d_bsc);
    else if (ev.P > 0)
        CUGO_LAUNCH_T(k_hsc_diag, S, dim3(xcd_grid(ev.P)), dim3(HD_BS), 0, s, ev,
                      hs.d_rowptr, damp_hsc_diag ? lambda : 0.0, d_Hpp, d_bp, d_bl, d_Hpl, (const S*)d_T,
                      d_Hsc, d_bsc);
}

void launch_schur(hipStream_t s, const cugo_edges& e, const cugo_hsc_struct& hs, double lambda,
                  int damp_hsc_diag, const double* d_Hpp, const double* d_bp, const double* d_Hll,
                  const double* d_bl, const void* d_Hpl, double* d_invHll, void* d_T,
                  double* d_bsc, double* d_Hsc, bool have_T, SchurRows rows)
{
    if (e.block_f32)
        launch_schur_t(s, e, hs, lambda, damp_hsc_diag, d_Hpp, d_bp, d_Hll, d_bl,
                       static_cast<const float*>(d_Hpl), d_invHll, static_cast<float*>(d_T), d_bsc, d_Hsc, have_T, rows);
    else
        launch_schur_t(s, e, hs, lambda, damp_hsc_diag, d_Hpp, d_bp, d_Hll, d_bl,
                       static_cast<const double*>(d_Hpl), d_invHll, static_cast<double*>(d_T), d_bsc, d_Hsc, have_T, rows);
}

bool schur_rows_usable(const cugo_hsc_struct& hs, int max_row_nnz)
{
    const size_t lds = (36 * (size_t)std::max(max_row_nnz, 1) + hr_fixed_doubles()) * sizeof(double);
    return hs.d_rowptr && hs.d_colind && lds <= 150 * 1024;
}

void launch_list_pos(hipStream_t s, const cugo_edges& e, int n_list, size_t M, const int32_t* d_off_ei,
                     int32_t* d_pose_pos, int32_t* d_off_pi)
{
    const EV ev = make_ev(e);
    if (n_list > 0)
        CUGO_LAUNCH(k_pose_pos, dim3(div_up(n_list, BS)), dim3(BS), 0, s, ev, n_list, d_pose_pos);
    if (M > 0)
        CUGO_LAUNCH(k_list_pos, dim3((unsigned)((M + BS - 1) / BS)), dim3(BS), 0, s, M, d_off_ei, (const int32_t*)d_pose_pos,
                    d_off_pi);
}

void launch_pose_rec(hipStream_t s, const cugo_edges& e, int n, int32_t* d_rec)
{
    const EV ev = make_ev(e);
    if (n > 0)
        CUGO_LAUNCH(k_pose_rec, dim3(div_up(n, BS)), dim3(BS), 0, s, ev, n, reinterpret_cast<int4*>(d_rec));
}

template <typename S>
static int launch_backsubst_update_t(hipStream_t s, const cugo_edges& e, double lambda, double lambda_pose,
                                      const double* d_invHll, const double* d_bl, const double* d_bp,
                                      const S* d_Hpl, const double* d_xp, double* d_xl,
                                      const double* d_poses_in, const double* d_lms_in, double* d_poses_out,
                                      double* d_lms_out, ReduceScratch rs, double* d_scale, const double* d_lmrec)
{
    const EV ev = make_ev(e);
    const int nbl = div_up(ev.E > ev.L ? ev.E : ev.L, BS), nbp = div_up(ev.P, BS);
    if (nbl + nbp > 0)
        CUGO_LAUNCH_T(k_backsubst_landmarks, S, dim3(nbl + nbp), dim3(BS), 0, s, ev, lambda, d_invHll,
                      d_bl, d_Hpl, d_xp, d_xl, d_lms_in, d_lms_out, rs.d_partials, nbl,
                      lambda_pose, d_bp, d_poses_in, d_poses_out, d_lmrec);
    if (d_scale) // nullptr: the partials stay in the scratch for launch_errors_tail
        CUGO_LAUNCH(k_sum_partials, dim3(1), dim3(SP_BS), 0, s, rs.d_partials, nbl + nbp, d_scale);
    return nbl + nbp;
}

int launch_backsubst_update(hipStream_t s, const cugo_edges& e, double lambda, double lambda_pose,
                            const double* d_invHll, const double* d_bl, const double* d_bp,
                            const void* d_Hpl, const double* d_xp, double* d_xl,
                            const double* d_poses_in, const double* d_lms_in, double* d_poses_out,
                            double* d_lms_out, ReduceScratch rs, double* d_scale, const double* d_lmrec)
{
    if (e.block_f32)
        return launch_backsubst_update_t(s, e, lambda, lambda_pose, d_invHll, d_bl, d_bp, static_cast<const float*>(d_Hpl),
                                  d_xp, d_xl, d_poses_in, d_lms_in, d_poses_out, d_lms_out, rs, d_scale, d_lmrec);
    return launch_backsubst_update_t(s, e, lambda, lambda_pose, d_invHll, d_bl, d_bp, static_cast<const double*>(d_Hpl),
                                  d_xp, d_xl, d_poses_in, d_lms_in, d_poses_out, d_lms_out, rs, d_scale, d_lmrec);
}

} // namespace cugo_k
