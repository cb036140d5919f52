// Engine implementation: flatten -> device, structure build, LM loop.
// ref: src/block_solver.cpp (step order), src/cuda_graph_optimisation.cpp:48-154 (LM control),
//      src/sparse_block_matrix.cpp:63-156 (Hsc pattern), src/optimisable_graph.hpp:642-661.
#include "engine.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <numeric>
#include <thread>

#include "../kernels/kernels.h"
#include "chol_solver.h"
#include "hip_util.h"
#include "options.h"
#include "rccl_comm.h"
#include "schur_plan.h"
#include "structure_gpu.h"
#include "thread_pool.h"

namespace cugo_host
{

namespace
{
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#endif
}
thread_local std::string g_last_error;
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
}
} // namespace

void shard_range(const std::vector<int32_t>& lm_cnt, int rank, int world, int& l0, int& l1)
{
    const int Lall = (int)lm_cnt.size() - 1;
    l0 = 0, l1 = Lall;
    if (world <= 1)
        return;
    const int64_t Etot = lm_cnt[Lall];
    auto cut = [&](int r) {
        const int64_t target = Etot * r / world;
        return (int)(std::lower_bound(lm_cnt.begin(), lm_cnt.end(), (int32_t)target) - lm_cnt.begin());
    };
    l0 = rank == 0 ? 0 : std::min(cut(rank), Lall);
    l1 = rank == world - 1 ? Lall : std::min(cut(rank + 1), Lall);
    if (l1 < l0)
        l1 = l0;
}

void set_last_error(const std::string& s) { g_last_error = s; }
const char* get_last_error() { return g_last_error.c_str(); }

struct Engine::Impl : cugo_k::LaunchHook
{
    cugo_ctx ctx;
    cugo_chol chol;
    int rank = 0, world = 1;
    int init_rank = -1, init_world = -1; // what the last full initialize() was built for
    cugo_exchange_fn xfn = nullptr;
    void* xuser = nullptr;
    std::shared_ptr<RcclComm> comm; // native exchange (RCCL on the solver's stream); else xfn
    double xchg_bytes = 0;          // payload bytes all-reduced since initialize()
    int xchg_calls = 0;
    bool profile = false;
    bool plan_only = false; // host side only (no device): see Engine::Engine
    Options opt = Options::from_env(); // the environment switches, read once when the optimiser is created

    int Pall = 0, Lall = 0, P = 0, L = 0, E = 0;
    int shard_l0 = 0, shard_l1 = 0; // landmark index range owned by this rank
    cugo_robust rk{CUGO_RK_NONE, 1.0, CUGO_RK_NONE, 1.0};
    int n_omega = 1, n_cams = 1;

    // host copies (sorted, local shard)
    std::vector<int32_t> h_e_pose, h_e_lm, h_lm_ptr, h_pose_ptr, h_pose_edge;
    std::vector<int32_t> slot_edge;     // slot -> edge index of the FlatGraph (-1: padding)
    // staging of initialize(), kept between calls: a fresh 40 MB of vectors costs more in page
    // faults than the work done on them
    FlatGraph staging;
    bool f32_blocks = false; // Hpl / T stored as float (fp32-internal mode)
    std::vector<int32_t> st_lm_cnt, st_order, st_slot_src;
    std::vector<double> st_meas, st_omega;
    std::vector<uint16_t> st_cam;
    std::vector<double> slot_threshold; // slot -> outlier threshold (empty: rejection disabled)
    int last_err_buf = 0;               // estimate buffer of the last error pass
    bool lm_in_one_group = false;       // no landmark's slots straddle two 256-slot groups
    int Etot = 0;
    std::vector<uint8_t> h_flags;
    // global co-visibility: free landmark -> sorted free poses (free-free active edges, all shards)
    std::vector<int32_t> cov_ptr, cov_pose;
    // Hsc pattern (host)
    std::vector<int32_t> hsc_rowptr, hsc_colind;

    DevBuf<int32_t> d_e_pose, d_e_lm, d_lm_ptr, d_pose_ptr, d_pose_edge;
    DevBuf<int32_t> d_pose_rec; // [n][4] per entry of the pose-major list: slot, end of its landmark, landmark, flags (k_hsc_rows)
    DevBuf<unsigned long long> d_hash; // diagnosis (CUGO_DEBUG_HASH=file): per iteration 16 stage checksums
    uint64_t trial_seq = 0;        // sequence number of the last LM trial whose result the host waited for
    hipEvent_t trial_ev = nullptr; // end of a trial whose successor build is already queued (optimize)
    bool rows_on = false;       // the Schur complement by block rows (k_hsc_rows; CUGO_HSC_ROWS=0: the gather kernels)
    DevBuf<int32_t> d_pose_pos, d_off_pi; // row-strip form of the off-diagonal gather (k_hsc_offdiag_strip)
    bool strip_on = false;
    int max_row_nnz = 0;
    DevBuf<double> d_meas, d_omega, d_cams;
    DevBuf<uint8_t> d_flags;
    DevBuf<uint16_t> d_cam;
    DevBuf<double> d_poses[2], d_lms[2];
    int cur = 0;
    DevBuf<double> d_Hpp, d_b, d_Hll, d_Hpl, d_T, d_invHll, d_x, d_sys, d_tmp, d_scal;
    // fused iteration (Options::pose_schur): {L^-1, L^-1 bl} of every landmark (Hll + lambda I = L L^T) in a 128-byte
    // slot for k_pose_schur / k_backsubst_landmarks, and whether Hpp, Hpl and invHll are those of the current
    // linearisation: a fused build pass writes ONE block stream (G = Hpl L^-T, into d_T) and none of the three
    DevBuf<double> d_lmrec;
    bool hpp_valid = true;
    PinnedBuf<double> h_pin_poses, h_pin_lms; // refresh_estimates_pinned / download_pinned
    Clock::duration last_trial_wait{0};       // how long the host polled for the previous trial's result
    DevBuf<int32_t> d_hsc_rowptr, d_hsc_colind, d_off_ptr, d_off_ei, d_off_ej, d_fail;
    PinnedBuf<double> h_scal;
    PinnedBuf<int32_t> h_fail;

    cugo_edges ev{};
    cugo_hsc_struct hs{};
    SchurPlanDevice splan; // landmark-major product plan of the Schur complement (schur_plan.h)
    bool splan_on = false;
    GpuStructure gstruct;  // Hsc pattern + contribution lists built on the device (structure_gpu.h)
    bool structure_dirty = true;
    bool sig_valid = false; // sig_* hold the topology the structure at hand was built from
    // ... and the topology itself (compared on a hash hit), saved by build_structure()
    int sig_dims[8] = {0}, pending_dims[8] = {0};
    std::vector<int32_t> sig_e_pose, sig_e_lm, sig_cov_pose;
    // The Hsc pattern, the ordering and the symbolic factor depend on the co-visibility lists alone.  When
    // those change, initialize() starts their build on a helper thread and a second stream the moment the
    // lists exist — the slot layout, the slot arrays, their upload and the topology hash then run beside
    // it — and build_structure() (first optimize()) joins it, builds the contribution lists and uploads
    // the plan.  CUGO_ASYNC_STRUCTURE=0: everything in build_structure(), as before.
    std::thread pat_thread;
    std::exception_ptr pat_err;
    bool pat_async_ok = false;   // the helper finished a pattern + host plan for the current lists
    bool pattern_dirty = true;   // the pattern / plan in gstruct / chol are not those of the current lists
    bool plan_uploaded = false;
    // the helper's progress, for the caller of initialize(): 1 = the pattern is in gstruct (the contribution lists
    // can be built as soon as the slots are on the device), -1 = it was not built; and whether the lists of the
    // structure at hand were already built at the end of initialize()
    std::atomic<int> pat_stage{0};
    bool lists_built = false;
    int pat_P = -1, pat_L = -1;
    std::vector<int32_t> pat_cov_ptr, pat_cov_pose; // the lists the pattern in gstruct / chol was built from
    hipStream_t s2 = nullptr;
    void join_pattern()
    {
        if (pat_thread.joinable())
            pat_thread.join();
    }
    std::vector<uint8_t> sig_flags;

    // optional HIP-event timing: 1 = an event pair round every kernel group AND every kernel (each pair adds a few
    // microseconds to what it brackets: per-kernel figures, never sums), 2 = ONE event at every group boundary —
    // the group times then add up exactly to the device time between the first and the last event of optimize()
    int ktiming = 0;
    struct Mark
    {
        int label;
        hipEvent_t e;
    };
    std::vector<Mark> marks; // (mode 2)
    void mark(int label)
    {
        Mark mk{label, nullptr};
        CUGO_HIP(hipEventCreate(&mk.e));
        CUGO_HIP(hipEventRecord(mk.e, ctx.stream));
        marks.push_back(mk);
    }
    struct KEv
    {
        int label;
        hipEvent_t a, b;
    };
    std::vector<KEv> kev;
    std::vector<std::string> klabels;
    std::vector<double> kms;
    std::vector<int> kcount;
    int klabel(const char* name)
    {
        for (size_t i = 0; i < klabels.size(); i++)
            if (klabels[i] == name)
                return (int)i;
        klabels.push_back(name);
        kms.push_back(0);
        kcount.push_back(0);
        return (int)klabels.size() - 1;
    }
    template <typename F>
    void timed(const char* name, F&& f)
    {
        if (!ktiming)
        {
            f();
            return;
        }
        if (ktiming == 2)
        {
            mark(klabel(name));
            f();
            return;
        }
        KEv e;
        e.label = klabel(name);
        CUGO_HIP(hipEventCreate(&e.a));
        CUGO_HIP(hipEventCreate(&e.b));
        CUGO_HIP(hipEventRecord(e.a, ctx.stream));
        f();
        CUGO_HIP(hipEventRecord(e.b, ctx.stream));
        kev.push_back(e);
    }
    // per-kernel events (LaunchHook): nested inside the group scopes above
    std::vector<KEv> kev_open;
    void begin(const char* kernel, hipStream_t st) override
    {
        KEv e;
        e.label = klabel(kernel);
        CUGO_HIP(hipEventCreate(&e.a));
        CUGO_HIP(hipEventCreate(&e.b));
        CUGO_HIP(hipEventRecord(e.a, st));
        kev_open.push_back(e);
    }
    void end(const char*, hipStream_t st) override
    {
        KEv e = kev_open.back();
        kev_open.pop_back();
        CUGO_HIP(hipEventRecord(e.b, st));
        kev.push_back(e);
    }
    void collect_times()
    {
        if (!marks.empty())
        { // mode 2: the time from a group's mark to the next mark belongs to the group
            mark(-1);
            CUGO_HIP(hipStreamSynchronize(ctx.stream));
            for (size_t i = 0; i + 1 < marks.size(); i++)
            {
                float ms = 0;
                CUGO_HIP(hipEventElapsedTime(&ms, marks[i].e, marks[i + 1].e));
                kms[marks[i].label] += ms;
                kcount[marks[i].label]++;
            }
            for (auto& mk : marks)
                (void)hipEventDestroy(mk.e);
            marks.clear();
        }
        if (kev.empty())
            return;
        CUGO_HIP(hipStreamSynchronize(ctx.stream));
        for (auto& e : kev)
        {
            float ms = 0;
            CUGO_HIP(hipEventElapsedTime(&ms, e.a, e.b));
            kms[e.label] += ms;
            kcount[e.label]++;
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        kev.clear();
    }

    double* bp() { return d_b.data(); }
    double* bl() { return d_b.data() + 6 * (size_t)P; }
    double* xp() { return d_x.data(); }
    double* xl() { return d_x.data() + 6 * (size_t)P; }
    double* Hsc() { return d_sys.data(); }
    double* bsc() { return d_sys.data() + 36 * (size_t)hs.n_blocks; }
    cugo_k::ReduceScratch rs() { return {ctx.scratch.data(), ctx.scratch.size()}; }

    // all-reduce of n doubles at d over the shards.  With a communicator (cugo_graph_set_comm) this
    // is one ncclAllReduce queued on the solver's stream: no host synchronisation, no callback.
    // The callback form (cugo_graph_set_shard) is the test hook: it has to wait for the stream.
    void exchange(double* d, size_t n, int op)
    {
        if (world <= 1 && !comm) // a 1-rank communicator still issues its (identity) collectives
            return;
        xchg_bytes += 8.0 * (double)n;
        xchg_calls++;
        if (comm)
        {
            timed("exchange", [&] { comm->all_reduce(d, n, op, ctx.stream); });
            return;
        }
        if (!xfn)
            throw std::runtime_error("cugo: sharded run without a communicator or an exchange function");
        CUGO_HIP(hipStreamSynchronize(ctx.stream));
        xfn(d, n, op, xuser);
    }
    // the sparse LL^T of a sharded run: each rank factors the elimination subtrees it owns and the replicated
    // top of the tree (chol_symbolic: CholPlan::owner) and gets what crosses the boundary by broadcast
    void bind_solver_exchange()
    {
        join_pattern(); // (the helper's analyze_host() reads chol.rank / world / bcast)
        chol.rank = rank, chol.world = world;
        xs_ready = false;
        if (world > 1 || comm) // (a 1-rank communicator can run the owned form too: CUGO_OWN_SUBTREES=1, the rehearsal)
        {
            chol.bcast = [this](double* d, size_t n, int root) { broadcast(d, n, root); };
            chol.bcast_group = [this](bool start) {
                if (comm)
                    comm->group(start);
            };
        }
        else
            chol.bcast = nullptr, chol.bcast_group = nullptr;
        pattern_dirty = true; // the plan depends on (rank, world)
    }
    // ---- the per-trial exchange of the Schur system [Hsc | bsc] --------------------------------------------------
    // Replicated factorisation: every rank needs all of it — one sum all-reduce.  Rank-owned elimination subtrees
    // (chol_symbolic.h: CholPlan::owner): a rank assembles only the blocks and right-hand-side rows of its own
    // fronts and of the replicated top, so the system is packed by ownership (CholPlan::xs_off), the `world`
    // segments are REDUCE-SCATTERED (rank r receives the sum of segment r: 1 / world of the bytes) and the top's
    // part is all-reduced (reduce-scatter + all-gather of the top's rows), then the received units go back to
    // their places in d_sys.  CUGO_REDUCE_SCATTER=0: the all-reduce also in the owned form.
    DevBuf<int64_t> d_xs_off;
    DevBuf<double> d_xbuf;
    bool xs_ready = false; // d_xs_off / d_xbuf are those of the plan at hand
    void prepare_owned_exchange(hipStream_t s)
    {
        xs_ready = false;
        const CholPlan& pl = chol.plan;
        if (!chol.own_subtrees() || !opt.reduce_scatter || pl.xs_off.empty())
            return;
        d_xs_off.upload(pl.xs_off, s);
        d_xbuf.resize((size_t)(pl.xs_seg * world + pl.xs_top) + 16);
        d_xbuf.zero(s); // (the padding behind the shorter segments travels, and is never read)
        CUGO_HIP(hipStreamSynchronize(s));
        xs_ready = true;
    }
    void reduce_scatter(double* d, size_t n_seg)
    {
        xchg_bytes += 8.0 * (double)n_seg; // (bytes this rank receives)
        xchg_calls++;
        if (comm)
        {
            timed("exchange", [&] { comm->reduce_scatter(d, n_seg, ctx.stream); });
            return;
        }
        if (!xfn)
            throw std::runtime_error("cugo: sharded run without a communicator or an exchange function");
        CUGO_HIP(hipStreamSynchronize(ctx.stream));
        xfn(d, n_seg * (size_t)world, -1, xuser);
    }
    void exchange_system()
    {
        const size_t n = 36 * (size_t)hs.n_blocks + 6 * (size_t)P;
        if (!xs_ready)
        {
            exchange(d_sys.data(), n, 0);
            return;
        }
        const CholPlan& pl = chol.plan;
        const int B = hs.n_blocks;
        cugo_k::launch_xs_pack(ctx.stream, d_xs_off.data(), B, P, d_sys.data(), d_xbuf.data());
        reduce_scatter(d_xbuf.data(), (size_t)pl.xs_seg);
        if (pl.xs_top > 0)
            exchange(d_xbuf.data() + pl.xs_seg * world, (size_t)pl.xs_top, 0);
        cugo_k::launch_xs_unpack(ctx.stream, d_xs_off.data(), B, P, d_xbuf.data(), d_sys.data(), (long)(pl.xs_seg * rank),
                                 (long)(pl.xs_seg * (rank + 1)), (long)(pl.xs_seg * world));
    }
    // broadcast of n doubles at d from rank `root` (update blocks / solution ranges of rank-owned elimination
    // subtrees).  Callback form: op code 2 + root.
    double bcast_bytes = 0;
    int bcast_calls = 0;
    void broadcast(double* d, size_t n, int root)
    {
        if (world <= 1 && !comm)
            return;
        bcast_bytes += 8.0 * (double)n;
        bcast_calls++;
        if (comm)
        {
            timed("exchange", [&] { comm->broadcast(d, n, root, ctx.stream); });
            return;
        }
        if (!xfn)
            throw std::runtime_error("cugo: sharded run without a communicator or an exchange function");
        CUGO_HIP(hipStreamSynchronize(ctx.stream));
        xfn(d, n, 2 + root, xuser);
    }
};

const char* Engine::profile_name(int i)
{
    static const char* names[PROF_COUNT] = {"0: Initialize Optimizer", "1: Build Structure",
                                            "2: Compute Error",        "3: Build System",
                                            "4: Schur Complement",     "5: Symbolic Decomposition",
                                            "6: Numerical Decomposition", "7: Update Solution"};
    return names[i];
}

Engine::Engine(bool plan_only) : impl_(new Impl)
{
    if (plan_only)
    {
        impl_->plan_only = true;
        impl_->chol.ctx = nullptr; // host-only analysis (cugo_chol with a null context)
        return;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    {
        delete impl_;
        impl_ = nullptr;
        throw std::runtime_error("cugo: no HIP device available (there is no CPU fallback)");
    }
    int dev = 0;
    CUGO_HIP(hipGetDevice(&dev));
    impl_->ctx.device = dev;
    impl_->ctx.stream = cache_stream_acquire();
    impl_->chol.ctx = &impl_->ctx;
    impl_->profile = impl_->opt.profile;
}

Engine::~Engine()
{
    if (impl_)
    {
        if (cugo_k::launch_hook() == impl_)
            cugo_k::set_launch_hook(nullptr);
        impl_->join_pattern();
        hipStream_t s = impl_->ctx.stream, s2 = impl_->s2;
        if (s)
            (void)hipStreamSynchronize(s);
        if (s2)
            (void)hipStreamSynchronize(s2);
        if (impl_->trial_ev)
            (void)hipEventDestroy(impl_->trial_ev);
        delete impl_;
        cache_stream_release(s); // back to the process-wide pool: creating one costs 1-2 ms
        if (s2)
            cache_stream_release(s2);
    }
}

Options& Engine::options() { return impl_->opt; }

void Engine::set_kernel_timing(int mode)
{
    impl_->ktiming = mode;
    cugo_k::set_launch_hook(mode == 1 ? impl_ : nullptr);
}

std::vector<Engine::KernelTime> Engine::kernel_times() const
{
    std::vector<KernelTime> out;
    for (size_t i = 0; i < impl_->klabels.size(); i++)
        out.push_back({impl_->klabels[i], impl_->kms[i], impl_->kcount[i]});
    return out;
}

void Engine::set_shard(int rank, int world, cugo_exchange_fn fn, void* user)
{
    if (world < 1 || rank < 0 || rank >= world)
        throw std::runtime_error("cugo: bad shard");
    impl_->comm.reset();
    impl_->rank = rank, impl_->world = world, impl_->xfn = fn, impl_->xuser = user;
    impl_->bind_solver_exchange();
}

void Engine::set_comm(std::shared_ptr<RcclComm> comm)
{
    if (!comm)
        throw std::runtime_error("cugo: null communicator");
    impl_->rank = comm->rank(), impl_->world = comm->world(), impl_->xfn = nullptr, impl_->xuser = nullptr;
    impl_->comm = std::move(comm);
    impl_->bind_solver_exchange();
}

void Engine::exchange_stats(double& bytes, int& calls) const
{
    bytes = impl_->xchg_bytes, calls = impl_->xchg_calls;
}


// CUGO_INIT_TIMING=1: per-section host times of initialize() on stderr (diagnosis only)
struct InitLaps
{
    bool on;
    explicit InitLaps(bool enabled) : on(enabled) {}
    Clock::time_point t = Clock::now();
    void lap(const char* what)
    {
        if (!on)
            return;
        const auto n = Clock::now();
        std::fprintf(stderr, "[cugo init] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

void Engine::set_float32_blocks(bool on)
{
    impl_->f32_blocks = on;
}

FlatGraph& Engine::staging()
{
    return impl_->staging;
}

void Engine::initialize(FlatGraph& g)
{
    const auto t0 = Clock::now();
    Impl& m = *impl_;
    InitLaps laps(m.opt.init_timing);
    // the helper thread of the previous initialize() reads m.P / m.L, the co-visibility lists and the solver's
    // (rank, world): nothing below may change them while it runs
    m.join_pattern();
    hipStream_t s = m.ctx.stream;
    m.Pall = g.Pall, m.Lall = g.Lall, m.P = g.P, m.L = g.L;
    m.rk = g.rk;
    m.init_rank = m.rank, m.init_world = m.world;
    const int Etot = g.n_edges();
    E_global_ = Etot;

    // ---- landmark-major order: counting sort by landmark, then by pose inside ----------
    // Threads own contiguous landmark ranges: each scans all edges (8 B per edge, from cache)
    // and counts / places only the edges of its own landmarks, so no two threads write the same
    // bin and the placement keeps the container order inside a landmark (stable).
    std::vector<int32_t>& lm_cnt = m.st_lm_cnt;
    std::vector<int32_t>& order = m.st_order;
    lm_cnt.assign(m.Lall + 1, 0);
    order.resize(Etot);
    {
        const int32_t* elm = g.e_lm.data();
        auto sort_by_pose = [&](size_t la, size_t lb) {
            for (size_t l = la; l < lb; l++)
            {
                int32_t* b = order.data() + lm_cnt[l];
                const int k = lm_cnt[l + 1] - lm_cnt[l];
                for (int i = 1; i < k; i++)
                { // insertion sort by pose index (k is small)
                    const int32_t v = b[i];
                    const int pv = g.e_pose[v];
                    int j = i - 1;
                    while (j >= 0 && g.e_pose[b[j]] > pv)
                    {
                        b[j + 1] = b[j];
                        j--;
                    }
                    b[j + 1] = v;
                }
            }
        };
        // Callers usually add the edges landmark by landmark (ORB-SLAM2 walks its map points):
        // then the counting sort is the identity and only the counts are needed.
        std::vector<uint8_t> chunk_sorted(pool_threads(), 1);
        parallel_chunks((size_t)Etot, 100000, [&](size_t a, size_t b, unsigned t) {
            bool ok = true;
            for (size_t e = std::max<size_t>(a, 1); e < b; e++)
                ok = ok && elm[e - 1] <= elm[e];
            chunk_sorted[t] = ok;
        });
        bool sorted = true;
        for (uint8_t c : chunk_sorted)
            sorted = sorted && c;
        if (sorted)
        {
            // a chunk's first and last landmark may continue in the neighbouring chunks: their
            // counts are combined afterwards
            struct Side
            {
                int32_t first = -1, nfirst = 0, last = -1, nlast = 0;
            };
            std::vector<Side> side(pool_threads());
            parallel_chunks((size_t)Etot, 100000, [&](size_t a, size_t b, unsigned t) {
                if (a == b)
                    return;
                Side sd;
                sd.first = elm[a], sd.last = elm[b - 1];
                for (size_t e = a; e < b; e++)
                {
                    const int32_t l = elm[e];
                    if (l == sd.first)
                        sd.nfirst++;
                    else if (l == sd.last)
                        sd.nlast++;
                    else
                        lm_cnt[l + 1]++;
                    order[e] = (int32_t)e;
                }
                side[t] = sd;
            });
            for (const Side& sd : side)
            {
                if (sd.first >= 0)
                    lm_cnt[sd.first + 1] += sd.nfirst;
                if (sd.last >= 0 && sd.last != sd.first)
                    lm_cnt[sd.last + 1] += sd.nlast;
            }
            for (int l = 0; l < m.Lall; l++)
                lm_cnt[l + 1] += lm_cnt[l];
            parallel_chunks((size_t)m.Lall, 65536, [&](size_t la, size_t lb, unsigned) { sort_by_pose(la, lb); });
        }
        else
        {
            // Not sorted as a whole — but callers add their edges set by set and, inside a set, landmark by
            // landmark (the reference sample: all mono edges, then all stereo edges, each in landmark order): the
            // container order is a FEW sorted runs.  Then the counting sort is a merge of runs: a thread owns a
            // landmark range, finds its part of every run by binary search and lays the edges of each landmark
            // down run after run (= container order: stable).  O(E) work in all, where the general path below
            // has every thread scan all edges twice.
            std::vector<int32_t> run_start(1, 0);
            {
                const unsigned ntd = pool_threads();
                std::vector<std::vector<int32_t>> desc(ntd);
                parallel_chunks((size_t)Etot, 100000, [&](size_t a, size_t b, unsigned t) {
                    for (size_t e = std::max<size_t>(a, 1); e < b && desc[t].size() <= 64; e++)
                        if (elm[e - 1] > elm[e])
                            desc[t].push_back((int32_t)e);
                });
                for (const auto& d : desc)
                    run_start.insert(run_start.end(), d.begin(), d.end());
            }
            const int nruns = (int)run_start.size();
            run_start.push_back(Etot);
            if (nruns <= 64)
            {
                const unsigned nt = std::max(1u, std::min<unsigned>(pool_threads(), (unsigned)std::max(1, m.Lall / 1024)));
                std::vector<int64_t> base(nt + 1, 0);
                auto lrange = [&](unsigned t) { return (size_t)m.Lall * t / nt; };
                auto run_lo = [&](int r, size_t l) {
                    return (int32_t)(std::lower_bound(elm + run_start[r], elm + run_start[r + 1], (int32_t)l) - elm);
                };
                struct MergeCtx
                {
                    const std::function<void(unsigned)>* fn;
                };
                auto run_threads = [&](const std::function<void(unsigned)>& fn) {
                    MergeCtx c{&fn};
                    pool_run(
                        nt, [](void* p, unsigned t) { (*static_cast<MergeCtx*>(p)->fn)(t); }, &c);
                };
                run_threads([&](unsigned t) {
                    int64_t c = 0;
                    for (int r = 0; r < nruns; r++)
                        c += run_lo(r, lrange(t + 1)) - run_lo(r, lrange(t));
                    base[t + 1] = c;
                });
                for (unsigned t = 0; t < nt; t++)
                    base[t + 1] += base[t];
                run_threads([&](unsigned t) {
                    const size_t la = lrange(t), lb = lrange(t + 1);
                    if (la == lb)
                        return;
                    std::vector<int32_t> cur(nruns), lim(nruns);
                    for (int r = 0; r < nruns; r++)
                        cur[r] = run_lo(r, la), lim[r] = run_lo(r, lb);
                    int32_t pos = (int32_t)base[t];
                    for (size_t l = la; l < lb; l++)
                    {
                        lm_cnt[l] = pos; // (start of landmark l; lm_cnt[Lall] is set below)
                        for (int r = 0; r < nruns; r++)
                            while (cur[r] < lim[r] && elm[cur[r]] == (int32_t)l)
                                order[pos++] = cur[r]++;
                    }
                });
                lm_cnt[m.Lall] = Etot;
                parallel_chunks((size_t)m.Lall, 65536, [&](size_t la, size_t lb, unsigned) { sort_by_pose(la, lb); });
            }
            else
            {
                parallel_chunks((size_t)m.Lall, 65536, [&](size_t la, size_t lb, unsigned) {
                    for (int e = 0; e < Etot; e++)
                    {
                        const size_t l = (size_t)elm[e];
                        if (l >= la && l < lb)
                            lm_cnt[l + 1]++;
                    }
                });
                for (int l = 0; l < m.Lall; l++)
                    lm_cnt[l + 1] += lm_cnt[l];
                parallel_chunks((size_t)m.Lall, 65536, [&](size_t la, size_t lb, unsigned) {
                    if (la == lb)
                        return;
                    std::vector<int32_t> pos(lm_cnt.begin() + la, lm_cnt.begin() + lb);
                    for (int e = 0; e < Etot; e++)
                    {
                        const size_t l = (size_t)elm[e];
                        if (l >= la && l < lb)
                            order[pos[l - la]++] = e;
                    }
                    sort_by_pose(la, lb);
                });
            }
        }
    }
    laps.lap("engine: landmark sort");
    // ---- global co-visibility (all shards): free landmark -> free poses ----------------
    {
        auto is_ff = [&](int e) {
            return (g.e_flags[e] & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE)) == 0;
        };
        m.cov_ptr.assign(m.L + 1, 0);
        parallel_chunks((size_t)m.L, 65536, [&](size_t la, size_t lb, unsigned) {
            for (size_t l = la; l < lb; l++)
            {
                int c = 0;
                for (int i = lm_cnt[l]; i < lm_cnt[l + 1]; i++)
                    c += is_ff(order[i]);
                m.cov_ptr[l + 1] = c;
            }
        });
        for (int l = 0; l < m.L; l++)
            m.cov_ptr[l + 1] += m.cov_ptr[l];
        m.cov_pose.resize((size_t)m.cov_ptr[m.L]);
        std::atomic<int> dup_lm{-1};
        parallel_chunks((size_t)m.L, 65536, [&](size_t la, size_t lb, unsigned) {
            for (size_t l = la; l < lb; l++)
            {
                int o = m.cov_ptr[l];
                for (int i = lm_cnt[l]; i < lm_cnt[l + 1]; i++)
                    if (is_ff(order[i]))
                    {
                        // (the slots of a landmark are sorted by pose: equal poses are neighbours)
                        if (o > m.cov_ptr[l] && m.cov_pose[o - 1] == g.e_pose[order[i]])
                            dup_lm.store((int)l, std::memory_order_relaxed);
                        m.cov_pose[o++] = g.e_pose[order[i]];
                    }
            }
        });
        // Two active edges between the same free pose and free landmark: the reference stores ONE Hpl block
        // per (pose, landmark) pair and forms only one of the two cross products of such a pair
        // (ref: .cu:1347-1378 iterates j >= i inside a column), i.e. it has no defined behaviour for them;
        // here every structure (Hpl slots, Hsc lists, the device structure build) assumes distinct pairs.
        if (dup_lm.load() >= 0)
            throw std::runtime_error("cugo: two active edges join the same free pose and free landmark (landmark index " +
                                     std::to_string(dup_lm.load()) + "): duplicate (pose, landmark) edges are not supported");
    }
    laps.lap("engine: co-visibility");
    bool helper_started = false;
    // ---- the co-visibility lists decide the Hsc pattern: if they changed, its build, the ordering and the
    // symbolic factorisation start NOW, beside the rest of this function (see Impl::pat_thread)
    m.join_pattern();
    if (m.pat_err)
    {
        m.pat_err = nullptr; // a failed helper of an earlier call: that structure was never used
        m.pattern_dirty = true;
    }
    {
        auto same = [](const auto& a, const auto& b) {
            return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(a[0])) == 0);
        };
        const bool pat_same = !m.pattern_dirty && m.opt.structure_reuse && m.P == m.pat_P &&
                              m.L == m.pat_L && same(m.cov_ptr, m.pat_cov_ptr) && same(m.cov_pose, m.pat_cov_pose);
        m.pat_async_ok = false;
        m.lists_built = false;
        m.pat_stage.store(0, std::memory_order_relaxed);
        if (!pat_same)
        {
            m.pattern_dirty = true;
            m.plan_uploaded = false;
            const bool device_build = !m.plan_only && !m.opt.schur_plan && !m.opt.host_structure;
            if (device_build && m.opt.async_structure && m.P > 0 && !m.cov_pose.empty())
            {
                if (!m.s2)
                    m.s2 = cache_stream_acquire();
                m.pat_thread = std::thread([this] {
                    Impl& mm = *impl_;
                    try
                    {
                        // (the current device is per thread: a rank of a multi-GPU job runs on LOCAL_RANK, not 0)
                        CUGO_HIP(hipSetDevice(mm.ctx.device));
                        const auto t0p = Clock::now();
                        if (!build_pattern_gpu(mm.s2, mm.P, mm.L, mm.cov_ptr.data(), mm.cov_pose.data(), mm.gstruct))
                        {
                            mm.pat_stage.store(-1, std::memory_order_release);
                            return;
                        }
                        mm.pat_stage.store(1, std::memory_order_release);
                        mm.hsc_rowptr = mm.gstruct.h_rowptr, mm.hsc_colind = mm.gstruct.h_colind;
                        prof_[PROF_BUILD_STRUCTURE] += ms_since(t0p);
                        const auto t1p = Clock::now();
                        mm.chol.analyze_host(mm.P, mm.hsc_rowptr.data(), mm.hsc_colind.data());
                        // ... and the plan goes to the device from here as well (second stream; upload() waits for
                        // its copies): the first optimize() then only builds the contribution lists
                        mm.chol.upload(mm.s2);
                        mm.plan_uploaded = true;
                        prof_[PROF_SYMBOLIC] += ms_since(t1p);
                        mm.pat_async_ok = true;
                    }
                    catch (...)
                    {
                        mm.pat_err = std::current_exception();
                        mm.pat_stage.store(-1, std::memory_order_release);
                    }
                });
                helper_started = true;
            }
        }
    }
    // ---- shard: contiguous landmark range balanced by edge count -----------------------
    int l0 = 0, l1 = m.Lall;
    shard_range(lm_cnt, m.rank, m.world, l0, l1);
    m.shard_l0 = l0, m.shard_l1 = l1;
    // Slot layout of this shard's edges: landmark-major, padded with inactive slots so that no
    // landmark with <= 256 edges straddles a 256-slot boundary (k_build_edges / the back-
    // substitution sum a landmark's edges inside one workgroup).  A padding slot belongs to
    // the landmark before it; slot_src[i] = index into `order`, or -1 for padding.
    constexpr int kBlock = 256;
    std::vector<int32_t>& slot_src = m.st_slot_src;
    m.h_lm_ptr.assign(m.Lall + 1, 0);
    {
        // first the start slot of every landmark (sequential: a padding decision moves everything
        // behind it), then the slots are filled per landmark range in parallel
        std::vector<int32_t> start((size_t)(l1 - l0) + 1, 0);
        int pos = 0, last_l = -1; // last_l: last landmark that owns slots
        for (int l = l0; l < l1; l++)
        {
            const int k = lm_cnt[l + 1] - lm_cnt[l];
            if (k > 0 && k <= kBlock && pos % kBlock + k > kBlock && last_l >= 0)
            {
                pos += kBlock - pos % kBlock;
                for (int q = last_l + 1; q <= l; q++)
                    m.h_lm_ptr[q] = pos; // the padding extends landmark last_l
            }
            m.h_lm_ptr[l] = pos;
            start[l - l0] = pos;
            pos += k;
            if (k > 0)
                last_l = l;
        }
        const int total = pos;
        slot_src.assign((size_t)total, -1);
        parallel_chunks((size_t)(l1 - l0), 65536, [&](size_t a, size_t b, unsigned) {
            for (size_t q = a; q < b; q++)
            {
                const int l = l0 + (int)q;
                int32_t* dst = slot_src.data() + start[q];
                for (int i = lm_cnt[l]; i < lm_cnt[l + 1]; i++)
                    *dst++ = i;
            }
        });
        for (int l = l1; l <= m.Lall; l++)
            m.h_lm_ptr[l] = total;
        // lm_ptr[l] for l < l0 stays 0; fix up the entries between padded landmarks
        for (int l = l0 + 1; l <= l1; l++)
            m.h_lm_ptr[l] = std::max(m.h_lm_ptr[l], m.h_lm_ptr[l - 1]);
    }
    // does every landmark's slot range lie inside one 256-slot group?  (not if a landmark has more
    // edges than that; the build pass may then not form T = Hpl invHll on the side, see optimize())
    m.lm_in_one_group = true;
    for (int l = 0; l < m.Lall && m.lm_in_one_group; l++)
        if (m.h_lm_ptr[l + 1] > m.h_lm_ptr[l] && m.h_lm_ptr[l] / kBlock != (m.h_lm_ptr[l + 1] - 1) / kBlock)
            m.lm_in_one_group = false;
    laps.lap("engine: slot layout");
    const int E = (int)slot_src.size();
    m.E = E;
    m.h_e_pose.resize(E), m.h_e_lm.resize(E), m.h_flags.resize(E);
    std::vector<double>&meas = m.st_meas, &omega = m.st_omega;
    std::vector<uint16_t>& cam = m.st_cam;
    meas.resize(3 * (size_t)E); // every slot is written by fill_slots
    omega.clear(), cam.clear();
    m.n_omega = g.e_omega.size() > 1 ? E : 1;
    m.n_cams = (int)(g.cams.size() / 5);
    if (m.n_omega > 1)
        omega.resize(E);
    else
        omega.assign(1, g.e_omega.empty() ? 1.0 : g.e_omega[0]);
    if (m.n_cams > 1)
        cam.resize(E);
    m.slot_edge.assign(E, -1);
    m.slot_threshold.clear();
    bool any_threshold = false;
    for (double t : g.e_outlier_threshold)
        any_threshold = any_threshold || t > 0.0;
    if (any_threshold)
        m.slot_threshold.assign(E, 0.0);
    m.Etot = Etot;
    m.last_err_buf = 0;
    m.xchg_bytes = 0, m.xchg_calls = 0;
    auto fill_slots = [&](int ia, int ib) {
        for (int i = ia; i < ib; i++)
        {
            if (slot_src[i] < 0)
            { // padding: inactive edge of the landmark of the nearest real slot before it
                int j = i - 1;
                while (slot_src[j] < 0)
                    j--;
                m.h_e_pose[i] = 0;
                m.h_e_lm[i] = g.e_lm[order[slot_src[j]]];
                m.h_flags[i] = CUGO_EDGE_INACTIVE;
                meas[i] = meas[(size_t)E + i] = meas[2 * (size_t)E + i] = 0.0;
                if (m.n_omega > 1)
                    omega[i] = 0.0;
                if (m.n_cams > 1)
                    cam[i] = 0;
                continue;
            }
            const int e = order[slot_src[i]];
            m.slot_edge[i] = e;
            if (any_threshold)
                m.slot_threshold[i] = g.e_outlier_threshold[e];
            m.h_e_pose[i] = g.e_pose[e];
            m.h_e_lm[i] = g.e_lm[e];
            m.h_flags[i] = g.e_flags[e];
            meas[i] = g.e_meas[3 * (size_t)e];
            meas[(size_t)E + i] = g.e_meas[3 * (size_t)e + 1];
            meas[2 * (size_t)E + i] = g.e_meas[3 * (size_t)e + 2];
            if (m.n_omega > 1)
                omega[i] = g.e_omega[e];
            if (m.n_cams > 1)
                cam[i] = g.e_cam[e];
        }
    };
    // independent per slot: split over a few host threads for big graphs
    parallel_chunks((size_t)E, 100000, [&](size_t a, size_t b, unsigned) { fill_slots((int)a, (int)b); });
    laps.lap("engine: fill slots");
    // ---- upload of the slot arrays and the estimates: copies from pageable memory block their caller, so a
    // helper thread issues them while this one builds the pose-major view and hashes the topology (a plan-only
    // engine has no device: it skips to the topology signature)
    std::exception_ptr up_err; // (declared before the Joiner: the thread may still assign to it while unwinding)
    struct Joiner
    {
        std::thread t;
        hipStream_t s = nullptr;
        bool wait_stream = false;
        ~Joiner()
        {
            if (t.joinable())
                t.join();
            if (wait_stream && s) // unwinding: the staging vectors of the copies in flight are about to go
                (void)hipStreamSynchronize(s);
        }
    } uploader;
    uploader.s = s, uploader.wait_stream = !m.plan_only;
    if (!m.plan_only)
    {
        m.d_e_pose.resize(m.h_e_pose.size()), m.d_e_lm.resize(m.h_e_lm.size()), m.d_flags.resize(m.h_flags.size());
        m.d_meas.resize(meas.size()), m.d_omega.resize(omega.size()), m.d_cams.resize(g.cams.size());
        if (m.n_cams > 1)
            m.d_cam.resize(cam.size());
        m.d_lm_ptr.resize(m.h_lm_ptr.size());
        for (int k = 0; k < 2; k++)
            m.d_poses[k].resize(g.poses.size()), m.d_lms[k].resize(g.lms.size());
        auto uploads = [&] {
            try
            {
                CUGO_HIP(hipSetDevice(m.ctx.device)); // (the current device is per thread)
                m.d_e_pose.upload(m.h_e_pose, s), m.d_e_lm.upload(m.h_e_lm, s), m.d_flags.upload(m.h_flags, s);
                m.d_lm_ptr.upload(m.h_lm_ptr, s);
                m.d_meas.upload(meas, s), m.d_omega.upload(omega, s), m.d_cams.upload(g.cams, s);
                if (m.n_cams > 1)
                    m.d_cam.upload(cam, s);
                for (int k = 0; k < 2; k++)
                {
                    m.d_poses[k].upload(g.poses, s);
                    m.d_lms[k].upload(g.lms, s);
                }
            }
            catch (...)
            {
                up_err = std::current_exception();
            }
        };
        if (!m.opt.upload_thread) // (CUGO_UPLOAD_THREAD=0: the copies are issued by this thread)
        {
            uploads();
            if (up_err)
                std::rethrow_exception(up_err);
        }
        else
            uploader.t = std::thread(uploads);
    }
    // ---- pose-major view (stable counting sort => ascending landmark inside a pose) ----
    // threads own slot ranges: a histogram per thread, then offsets per (pose, thread) in thread
    // order, so every thread places its own slots and the slot order inside a pose is kept
    m.h_pose_ptr.assign(m.Pall + 1, 0);
    m.h_pose_edge.assign(std::max(E, 1), 0);
    {
        const size_t serial_below = (size_t)m.Pall * pool_threads() > (size_t)E ? (size_t)E + 1 : 100000;
        const int32_t* ep = m.h_e_pose.data();
        const int32_t* src = slot_src.data();
        std::vector<std::vector<int32_t>> hist(pool_threads());
        const unsigned nt = parallel_chunks((size_t)E, serial_below, [&](size_t a, size_t b, unsigned t) {
            std::vector<int32_t>& h = hist[t];
            h.assign(m.Pall, 0);
            for (size_t i = a; i < b; i++)
                if (src[i] >= 0)
                    h[ep[i]]++;
        });
        // per pose: its total over the threads' histograms (parallel over poses), the prefix sum over poses, then the
        // first output position of every (pose, thread) — thread order inside a pose keeps the slot order
        parallel_chunks((size_t)m.Pall, 2048, [&](size_t qa, size_t qb, unsigned) {
            for (size_t q = qa; q < qb; q++)
            {
                int32_t c = 0;
                for (unsigned t = 0; t < nt; t++)
                    c += hist[t][q];
                m.h_pose_ptr[q + 1] = c;
            }
        });
        for (int q = 0; q < m.Pall; q++)
            m.h_pose_ptr[q + 1] += m.h_pose_ptr[q];
        parallel_chunks((size_t)m.Pall, 2048, [&](size_t qa, size_t qb, unsigned) {
            for (size_t q = qa; q < qb; q++)
            {
                int32_t run = m.h_pose_ptr[q];
                for (unsigned t = 0; t < nt; t++)
                {
                    const int32_t c = hist[t][q];
                    hist[t][q] = run; // first output position of thread t for pose q
                    run += c;
                }
            }
        });
        parallel_chunks((size_t)E, serial_below, [&](size_t a, size_t b, unsigned t) {
            std::vector<int32_t>& pos = hist[t];
            for (size_t i = a; i < b; i++)
                if (src[i] >= 0)
                    m.h_pose_edge[pos[ep[i]]++] = (int32_t)i;
        });
    }
    laps.lap("engine: pose-major view");
    // ---- upload (a plan-only engine has no device: it skips to the topology signature) -----
    if (!m.plan_only)
    {
    m.d_pose_ptr.upload(m.h_pose_ptr, s);
    m.d_pose_edge.upload(m.h_pose_edge, s);
    m.cur = 0;
    m.d_Hpp.resize(36 * (size_t)m.P + 16), m.d_b.resize(6 * (size_t)m.P + 3 * (size_t)m.L + 16);
    m.d_Hll.resize(9 * (size_t)m.L + 16), m.d_invHll.resize(9 * (size_t)m.L + 16);
    {
        // block streams: 18 doubles per edge slot, or 18 floats (= 9 doubles of storage)
        m.ev.block_f32 = (m.f32_blocks || m.opt.float32) ? 1 : 0;
        const size_t per_edge = m.ev.block_f32 ? 9 : 18;
        m.d_Hpl.resize(per_edge * (size_t)E + 16);
        m.d_T.release(); // allocated on demand (gather kernels only), see optimize()
    }
    m.d_x.resize(6 * (size_t)m.P + 3 * (size_t)m.L + 16);
    m.d_tmp.resize(36 * (size_t)m.P + 16);
    m.d_scal.resize(16), m.d_fail.resize(4), m.h_scal.resize(16), m.h_fail.resize(4);
    m.d_fail.zero(s); // [0] zero-pivot flag, [2] arrival counter of the trial's last launch
    m.h_scal[5] = -1.0;
    m.d_x.zero(s);
    m.ctx.scratch.resize(cugo_k::reduce_scratch_doubles(E, m.P, m.L));
    laps.lap("engine: enqueue uploads");
    }
    bool uploads_done = m.plan_only;
    auto finish_uploads = [&] {
        if (uploads_done)
            return;
        if (uploader.t.joinable())
            uploader.t.join();
        if (up_err)
            std::rethrow_exception(up_err);
        CUGO_HIP(hipStreamSynchronize(s)); // host staging vectors go out of scope
        uploader.wait_stream = false;
        uploads_done = true;
    };

    cugo_edges& ev = m.ev;
    ev.n_edges = E, ev.n_poses_total = m.Pall, ev.n_landmarks_total = m.Lall;
    ev.n_poses_free = m.P, ev.n_landmarks_free = m.L;
    ev.d_pose = m.d_e_pose.data(), ev.d_lm = m.d_e_lm.data(), ev.d_meas = m.d_meas.data();
    ev.d_omega = m.d_omega.data(), ev.n_omega = m.n_omega, ev.d_flags = m.d_flags.data();
    ev.d_cam = m.n_cams > 1 ? m.d_cam.data() : nullptr, ev.d_cams = m.d_cams.data();
    ev.n_cams = m.n_cams, ev.d_lm_ptr = m.d_lm_ptr.data(), ev.d_pose_ptr = m.d_pose_ptr.data();
    ev.d_pose_edge = m.d_pose_edge.data();
    {
        // opt-in (CUGO_HSC_ROWS=1): the Schur complement by whole block rows (k_hsc_rows).  Correct and
        // list-free, but measured slower than the gather pair (355 vs 152 us per Schur complement on the
        // kitti_00 shape): its per-product accumulation into LDS is a chain of dependent LDS round trips
        // 
        m.rows_on = !m.plan_only && m.opt.hsc_rows && !m.opt.schur_plan;
        if (m.rows_on)
        {
            finish_uploads();
            const int n = m.h_pose_ptr[m.Pall];
            m.d_pose_rec.resize(4 * (size_t)std::max(n, 1) + 16);
            cugo_k::launch_pose_rec(s, ev, n, m.d_pose_rec.data());
        }
    }
    // Re-use of the Hsc structure, ordering and symbolic factor across optimize() calls when
    // the topology is unchanged (ref: the isDirty logic of BlockSolver::buildStructure,
    // block_solver.cpp:151-216, which skips the rebuild for clean edge sets).  Decided by comparing the
    // flattened topology — slots, flags (a changed fixed flag is noticed too) and co-visibility lists — with
    // the copy the structure at hand was built from: a parallel memcmp that stops at the first difference (a new
    // graph costs next to nothing here; rounds 1-3 hashed all of it first, 0.4 ms of every full initialize()).
    {
        const int dims[8] = {m.Pall, m.Lall, m.P, m.L, E, m.rank, m.world, Etot};
        auto same = [](const auto& a, const auto& b) {
            if (a.size() != b.size())
                return false;
            const size_t bytes = a.size() * sizeof(a[0]);
            if (bytes == 0)
                return true;
            const char* pa = reinterpret_cast<const char*>(a.data());
            const char* pb = reinterpret_cast<const char*>(b.data());
            std::atomic<bool> eq{true};
            parallel_chunks(bytes, 1u << 18, [&](size_t lo, size_t hi, unsigned) {
                for (size_t o = lo; o < hi && eq.load(std::memory_order_relaxed); o += 1u << 16)
                    if (std::memcmp(pa + o, pb + o, std::min<size_t>(1u << 16, hi - o)) != 0)
                        eq.store(false, std::memory_order_relaxed);
            });
            return eq.load();
        };
        const bool hit = m.opt.structure_reuse && m.sig_valid && std::memcmp(dims, m.sig_dims, sizeof dims) == 0 &&
                         same(m.h_e_pose, m.sig_e_pose) && same(m.h_e_lm, m.sig_e_lm) &&
                         same(m.h_flags, m.sig_flags) && same(m.cov_pose, m.sig_cov_pose);
        if (!hit)
            m.structure_dirty = true;
        std::memcpy(m.pending_dims, dims, sizeof dims);
    }
    laps.lap("engine: topology compare");
    finish_uploads();
    laps.lap("engine: upload sync");
    // The contribution lists need the pattern (the helper's first 0.4 ms) and the slots on the device (just now):
    // they are built here, on this stream, while the helper is still ordering and analysing — off the critical
    // path of the first optimize()
    if (helper_started && m.structure_dirty && !m.plan_only)
    {
        while (m.pat_stage.load(std::memory_order_acquire) == 0)
            std::this_thread::yield();
        if (m.pat_stage.load(std::memory_order_acquire) == 1)
            m.lists_built = build_lists_gpu(s, m.E, m.P, m.Lall, m.d_e_pose.data(), m.d_flags.data(), m.d_lm_ptr.data(), m.gstruct);
        laps.lap("engine: contribution lists");
    }
    prof_[PROF_INITIALIZE] += ms_since(t0);
    if (m.plan_only && m.structure_dirty)
        build_structure(); // no optimize() will follow: the structure is all there is to do
}

void Engine::fill_structure_stats(int B, double products, double offdiag_products)
{
    Impl& m = *impl_;
    // remember what this structure was built from (Engine::initialize compares on a hash hit)
    std::memcpy(m.sig_dims, m.pending_dims, sizeof m.sig_dims);
    m.sig_valid = m.opt.structure_reuse; // (re-use switched off: nothing will ever be compared with these)
    if (m.sig_valid)
    {
        // (7 MB on the kitti_00 shape, on the path of a new graph's first optimize(): copied by the pool)
        auto copy = [](auto& dst, const auto& src) {
            dst.resize(src.size());
            const size_t bytes = src.size() * sizeof(src[0]);
            if (bytes == 0)
                return; // (an empty vector's data() may be null: not a valid memcpy argument)
            const char* from = reinterpret_cast<const char*>(src.data());
            char* to = reinterpret_cast<char*>(dst.data());
            parallel_chunks(bytes, 1u << 18, [&](size_t a, size_t b, unsigned) { std::memcpy(to + a, from + a, b - a); });
        };
        copy(m.sig_e_pose, m.h_e_pose), copy(m.sig_e_lm, m.h_e_lm), copy(m.sig_flags, m.h_flags);
        copy(m.sig_cov_pose, m.cov_pose);
    }
    // row strips, opt-in (CUGO_HSC_STRIP=1; default: one wave per block anywhere): per product the position of
    // its T edge in the edge list of its pose.  Bit for bit the gather kernel's sums with each T block read once,
    // but one 16-wave workgroup per CU cannot hide the gather latency the way 32 independent waves do:
    // measured 13.69 vs 11.45 ms per step (kitti_00 shape), 59.6 vs 41.2 ms (10k-pose graph)
    m.strip_on = false;
    {
        const size_t M = (size_t)offdiag_products;
        if (!m.plan_only && !m.splan_on && m.hs.d_off_ei && M > 0 && m.opt.hsc_strip)
        {
            m.d_pose_pos.resize((size_t)std::max(m.E, 1) + 16), m.d_off_pi.resize(M + 16);
            cugo_k::launch_list_pos(m.ctx.stream, m.ev, m.h_pose_ptr[m.Pall], M, m.hs.d_off_ei, m.d_pose_pos.data(),
                                    m.d_off_pi.data());
            m.strip_on = true;
        }
    }
    m.max_row_nnz = 0;
    for (size_t q = 0; q + 1 < m.hsc_rowptr.size(); q++)
        m.max_row_nnz = std::max(m.max_row_nnz, (int)(m.hsc_rowptr[q + 1] - m.hsc_rowptr[q]));
    sstats_.hsc_blocks = B;
    sstats_.products = products;
    sstats_.offdiag_products = offdiag_products;
    sstats_.nnzL = m.chol.plan.nnzL;
    sstats_.chol_flops = m.chol.plan.flops;
    sstats_.supernodes = m.chol.plan.n_super;
    sstats_.stages = m.chol.plan.n_stages;
    sstats_.front_bytes = 8.0 * (double)m.chol.plan.front_doubles;
    {
        const auto& pl = m.chol.plan;
        sstats_.chol_rank_flops = pl.rank_flops, sstats_.chol_top_flops = pl.top_flops;
        double bytes = 0;
        for (size_t k = 0; k < pl.xu_front.size(); k++)
        {
            const int f = pl.xu_front[k];
            const int64_t c0 = 6LL * pl.ncb[f], c1 = 6LL * pl.nb[f];
            if (c1 > c0)
                bytes += 8.0 * (double)((c1 - 1 - c0) * pl.ldf[f] + c1 + 1 - c0);
        }
        for (size_t k = 0; k < pl.xx_lo.size(); k++)
            bytes += 8.0 * 6.0 * (pl.xx_hi[k] - pl.xx_lo[k]);
        sstats_.chol_bcast_bytes = bytes, sstats_.chol_bcasts = (double)(pl.xu_front.size() + pl.xx_lo.size());
    }
    // sharded: what the per-trial exchange of the Schur system hands to this rank
    sstats_.xchg_sys_full_bytes = sstats_.xchg_sys_bytes = 0;
    if (m.world > 1 || m.comm)
    {
        if (!m.plan_only)
            m.prepare_owned_exchange(m.ctx.stream);
        const auto& pl = m.chol.plan;
        sstats_.xchg_sys_full_bytes = 8.0 * (36.0 * B + 6.0 * m.P);
        const bool keyed = m.plan_only ? (pl.owned && m.opt.reduce_scatter && !pl.xs_off.empty()) : m.xs_ready;
        sstats_.xchg_sys_bytes = keyed ? 8.0 * (double)(pl.xs_seg + pl.xs_top) : sstats_.xchg_sys_full_bytes;
    }
    sstats_.up_potrf_flops = m.chol.plan.up_potrf_flops;
    sstats_.up_trsm_flops = m.chol.plan.up_trsm_flops;
    sstats_.up_syrk_flops = m.chol.plan.up_syrk_flops;
    sstats_.up_ea_bytes = m.chol.plan.up_ea_bytes;
    sstats_.backward_bytes = m.chol.plan.backward_bytes;
}

void Engine::refresh_estimates(const FlatGraph& g)
{
    const auto t0 = Clock::now();
    Impl& m = *impl_;
    if ((int)(g.poses.size() / 7) != m.Pall || (int)(g.lms.size() / 3) != m.Lall)
        throw std::runtime_error("cugo: refresh_estimates on another graph");
    if (m.rank != m.init_rank || m.world != m.init_world)
        throw std::runtime_error("cugo: refresh_estimates after the shard changed (a full initialize() is needed)");
    m.last_err_buf = 0;
    m.xchg_bytes = 0, m.xchg_calls = 0;
    m.cur = 0;
    if (!m.plan_only)
    {
        hipStream_t s = m.ctx.stream;
        // one copy from (pageable) host memory per array; the second buffer is filled on the device
        m.d_poses[0].upload(g.poses, s);
        m.d_lms[0].upload(g.lms, s);
        m.d_poses[1].resize(g.poses.size()), m.d_lms[1].resize(g.lms.size());
        if (!g.poses.empty())
            CUGO_HIP(hipMemcpyAsync(m.d_poses[1].data(), m.d_poses[0].data(), g.poses.size() * sizeof(double),
                                    hipMemcpyDeviceToDevice, s));
        if (!g.lms.empty())
            CUGO_HIP(hipMemcpyAsync(m.d_lms[1].data(), m.d_lms[0].data(), g.lms.size() * sizeof(double),
                                    hipMemcpyDeviceToDevice, s));
        m.d_x.zero(s);
        CUGO_HIP(hipStreamSynchronize(s)); // the staging vectors may change after return
    }
    prof_[PROF_INITIALIZE] += ms_since(t0);
}

double* Engine::pinned_poses()
{
    Impl& m = *impl_;
    m.h_pin_poses.resize(7 * (size_t)std::max(m.Pall, 1));
    return m.h_pin_poses.data();
}
double* Engine::pinned_lms()
{
    Impl& m = *impl_;
    m.h_pin_lms.resize(3 * (size_t)std::max(m.Lall, 1));
    return m.h_pin_lms.data();
}

void Engine::refresh_estimates_pinned()
{
    const auto t0 = Clock::now();
    Impl& m = *impl_;
    if (m.rank != m.init_rank || m.world != m.init_world)
        throw std::runtime_error("cugo: refresh_estimates after the shard changed (a full initialize() is needed)");
    m.last_err_buf = 0;
    m.xchg_bytes = 0, m.xchg_calls = 0;
    m.cur = 0;
    if (!m.plan_only)
    {
        hipStream_t s = m.ctx.stream;
        const size_t np = 7 * (size_t)m.Pall, nl = 3 * (size_t)m.Lall;
        const double* hp = pinned_poses();
        const double* hl = pinned_lms();
        for (int b = 0; b < 2; b++)
            m.d_poses[b].resize(np), m.d_lms[b].resize(nl);
        if (np)
        {
            CUGO_HIP(hipMemcpyAsync(m.d_poses[0].data(), hp, np * sizeof(double), hipMemcpyHostToDevice, s));
            CUGO_HIP(hipMemcpyAsync(m.d_poses[1].data(), m.d_poses[0].data(), np * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        if (nl)
        {
            CUGO_HIP(hipMemcpyAsync(m.d_lms[0].data(), hl, nl * sizeof(double), hipMemcpyHostToDevice, s));
            CUGO_HIP(hipMemcpyAsync(m.d_lms[1].data(), m.d_lms[0].data(), nl * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        m.d_x.zero(s);
        // no wait: the staging is the engine's own pinned memory; whoever gathers into it again before these copies
        // have run queues a later copy of the whole array behind them (same stream), and that one is what counts
    }
    prof_[PROF_INITIALIZE] += ms_since(t0);
}

bool Engine::download_pinned(const double** poses, const double** lms)
{
    Impl& m = *impl_;
    if (m.world > 1 || m.plan_only)
        return false;
    hipStream_t s = m.ctx.stream;
    const size_t np = 7 * (size_t)m.Pall, nl = 3 * (size_t)m.Lall;
    double* hp = pinned_poses();
    double* hl = pinned_lms();
    if (nl)
        CUGO_HIP(hipMemcpyAsync(hl, m.d_lms[m.cur].data(), nl * sizeof(double), hipMemcpyDeviceToHost, s));
    if (np)
        CUGO_HIP(hipMemcpyAsync(hp, m.d_poses[m.cur].data(), np * sizeof(double), hipMemcpyDeviceToHost, s));
    CUGO_HIP(hipStreamSynchronize(s));
    *poses = hp, *lms = hl;
    return true;
}

// Hsc pattern from landmark co-visibility + contribution lists + Cholesky analysis
// (ref: buildStructure, block_solver.cpp:139-248)
void Engine::build_structure()
{
    Impl& m = *impl_;
    const auto t0 = Clock::now();
    InitLaps laps(m.opt.init_timing);
    hipStream_t s = m.ctx.stream;
    const int P = m.P, L = m.L;
    // ---- device-side build: pairs per landmark -> stable radix sort by pose pair -> runs = off-diagonal
    // blocks; the same lists, in the same order, as the host passes below (which remain for plan-only
    // and forced runs).  A shard builds the pattern from the global co-visibility lists.
    const bool shard = m.world > 1 || m.comm; // local slots are a subset: the pattern comes from the global lists
    m.join_pattern();
    if (m.pat_err)
    {
        std::exception_ptr e = m.pat_err;
        m.pat_err = nullptr;
        std::rethrow_exception(e);
    }
    if (!m.plan_only && (m.E > 0 || shard) && !m.opt.schur_plan && !m.opt.host_structure &&
        P > 0 && !m.cov_pose.empty())
    {
        // phase 1 (unless the helper thread of initialize() has done it, or the lists are those of the
        // pattern at hand): Hsc pattern from the co-visibility lists, ordering + symbolic factorisation
        bool ok = true;
        if (m.pattern_dirty && !m.pat_async_ok)
        {
            ok = build_pattern_gpu(s, P, L, m.cov_ptr.data(), m.cov_pose.data(), m.gstruct);
            if (ok)
            {
                m.hsc_rowptr = m.gstruct.h_rowptr, m.hsc_colind = m.gstruct.h_colind;
                prof_[PROF_BUILD_STRUCTURE] += ms_since(t0);
                const auto t1g = Clock::now();
                m.chol.analyze_host(P, m.hsc_rowptr.data(), m.hsc_colind.data());
                prof_[PROF_SYMBOLIC] += ms_since(t1g);
            }
        }
        laps.lap("structure: pattern + symbolic (unless done beside initialize)");
        // phase 2: the contribution lists from the slots on the device
        const auto t2 = Clock::now();
        const bool have_lists = m.lists_built && m.pat_async_ok; // (built at the end of initialize() for this pattern)
        m.lists_built = false;
        if (ok && (have_lists || build_lists_gpu(s, m.E, P, m.Lall, m.d_e_pose.data(), m.d_flags.data(), m.d_lm_ptr.data(), m.gstruct)))
        {
            laps.lap("structure: contribution lists (device)");
            if (!m.plan_uploaded) // (the helper thread of initialize() uploads the plans it analyses)
            {
                m.chol.upload(s);
                m.plan_uploaded = true;
            }
            if (m.pattern_dirty)
                m.pat_P = P, m.pat_L = L, m.pat_cov_ptr = m.cov_ptr, m.pat_cov_pose = m.cov_pose;
            m.pattern_dirty = false;
            m.gstruct.scratch.release(), m.gstruct.scratch2.release();
            const int B = m.gstruct.B;
            m.d_sys.resize(36 * (size_t)B + 6 * (size_t)P + 16);
            m.hs = cugo_hsc_struct{};
            m.hs.n_blocks = B;
            m.hs.d_rowptr = m.gstruct.rowptr.data(), m.hs.d_colind = m.gstruct.colind.data();
            m.hs.d_off_ptr = m.gstruct.off_ptr.data(), m.hs.d_off_ei = m.gstruct.off_ei.data();
            m.hs.d_off_ej = m.gstruct.off_ej.data();
            m.splan_on = false;
            double nff = 0;
            {
                std::vector<int64_t> part(pool_threads(), 0);
                parallel_chunks((size_t)m.E, 100000, [&](size_t a, size_t b, unsigned t) {
                    int64_t c = 0;
                    for (size_t e = a; e < b; e++)
                        c += (m.h_flags[e] & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE)) == 0;
                    part[t] = c;
                });
                for (int64_t c : part)
                    nff += (double)c;
            }
            prof_[PROF_BUILD_STRUCTURE] += ms_since(t2);
            laps.lap("structure: plan upload");
            // all products of the graph (every free-free edge also has its diagonal one) / local off-diagonal ones
            fill_structure_stats(B, (double)m.gstruct.Mglobal + (shard ? (double)m.cov_pose.size() : nff),
                                 (double)m.gstruct.Moff);
            sstats_.schur_slots = 0;
            m.structure_dirty = false;
            return;
        }
        m.pattern_dirty = true, m.plan_uploaded = false; // the host build below starts from scratch
    }
    m.pattern_dirty = true, m.plan_uploaded = false;
    // pose-major view of the global co-visibility
    std::vector<int32_t> pc_ptr(P + 1, 0), pc_lm(m.cov_pose.size());
    for (int32_t p : m.cov_pose)
        pc_ptr[p + 1]++;
    for (int p = 0; p < P; p++)
        pc_ptr[p + 1] += pc_ptr[p];
    {
        std::vector<int32_t> pos(pc_ptr.begin(), pc_ptr.end() - 1);
        for (int l = 0; l < L; l++)
            for (int k = m.cov_ptr[l]; k < m.cov_ptr[l + 1]; k++)
                pc_lm[pos[m.cov_pose[k]]++] = l;
    }
    // The three passes below are independent per pose row: contiguous row ranges, balanced by
    // their number of co-visibility entries, go to a few host threads (SLAM calls BA with a new
    // topology every time, so this "cold" work is paid on every call there).
    const unsigned nth = m.cov_pose.size() < 200000 ? 1u : std::max(1u, pool_threads());
    std::vector<int> row_split(nth + 1, P);
    row_split[0] = 0;
    for (unsigned t = 1; t < nth; t++)
    {
        const int32_t target = (int32_t)((int64_t)pc_ptr[P] * t / nth);
        row_split[t] = (int)(std::lower_bound(pc_ptr.begin(), pc_ptr.end(), target) - pc_ptr.begin());
        row_split[t] = std::min(std::max(row_split[t], row_split[t - 1]), P);
    }
    auto parallel_rows = [&](const std::function<void(unsigned, int, int)>& fn) {
        if (nth == 1)
            return fn(0, 0, P);
        struct RowCtx
        {
            const std::function<void(unsigned, int, int)>* fn;
            const std::vector<int>* split;
        } rc{&fn, &row_split};
        pool_run(
            nth,
            [](void* p, unsigned t) {
                RowCtx& c = *static_cast<RowCtx*>(p);
                (*c.fn)(t, (*c.split)[t], (*c.split)[t + 1]);
            },
            &rc);
    };
    laps.lap("structure: pose-major covis");
    // rows: diagonal first, then ascending columns (ref: sparse_block_matrix.cpp:80-155; O(M)
    // with a marker array instead of the reference's dense P x P byte map)
    m.hsc_rowptr.assign(P + 1, 0);
    std::vector<std::vector<int32_t>> cols_t(nth);
    std::vector<double> products_t(nth, 0.0);
    parallel_rows([&](unsigned t, int p0, int p1) {
        std::vector<int32_t> mark(P, -1);
        std::vector<int32_t>& cols = cols_t[t];
        double products = 0;
        for (int p = p0; p < p1; p++)
        {
            const size_t start = cols.size();
            cols.push_back(p);
            mark[p] = p;
            for (int i = pc_ptr[p]; i < pc_ptr[p + 1]; i++)
            {
                const int l = pc_lm[i];
                for (int k = m.cov_ptr[l]; k < m.cov_ptr[l + 1]; k++)
                {
                    const int q = m.cov_pose[k];
                    if (q >= p)
                        products += 1;
                    if (q > p && mark[q] != p)
                    {
                        mark[q] = p;
                        cols.push_back(q);
                    }
                }
            }
            std::sort(cols.begin() + start + 1, cols.end());
            m.hsc_rowptr[p + 1] = (int32_t)(cols.size() - start); // row length; prefix sum below
        }
        products_t[t] = products;
    });
    double products = 0;
    m.hsc_colind.clear();
    for (unsigned t = 0; t < nth; t++)
    {
        products += products_t[t];
        m.hsc_colind.insert(m.hsc_colind.end(), cols_t[t].begin(), cols_t[t].end());
    }
    for (int p = 0; p < P; p++)
        m.hsc_rowptr[p + 1] += m.hsc_rowptr[p];
    const int B = (int)m.hsc_colind.size();
    laps.lap("structure: Hsc pattern");
    // Landmark-major product plan (schur_plan.h), on request (CUGO_SCHUR_PLAN=1): every Hpl block is
    // then read once per Schur complement and T is never written (284 MB instead of 730 MB of
    // memory-side traffic on the kitti_00 shape) — but the kernel pair takes as long as the
    // destination-major gather kernels (196 vs 192 us there: half of its partial slots hold a single
    // product) and the plan costs 4 ms more to build, so the gather kernels stay the default
    // (DESIGN.md section 4b).  Unusable if a landmark's active edges straddle two 256-slot groups.
    SchurPlanHost sp;
    if (m.opt.schur_plan)
        build_schur_plan(m.E, P, m.h_e_pose.data(), m.h_e_lm.data(), m.h_flags.data(), m.hsc_rowptr.data(),
                         m.hsc_colind.data(), sp);
    m.splan_on = sp.usable;
    laps.lap("structure: landmark-major product plan");
    // contribution lists of the off-diagonal blocks from the LOCAL edges, built row by row
    // (pose-major): pos[q] gives the slot of column q in the current row, so every product is
    // placed with O(1) work; inside a block the contributions are in ascending landmark order.
    // A row only touches the counters / slots of its own blocks: rows are independent.
    auto free_free = [&](int e) {
        return (m.h_flags[e] & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE)) == 0;
    };
    std::vector<int32_t> off_cnt(B + 1, 0);
    if (!m.splan_on)
    parallel_rows([&](unsigned, int p0, int p1) {
        std::vector<int32_t> pos(P, -1);
        for (int p = p0; p < p1; p++)
        {
            const int r0 = m.hsc_rowptr[p], r1 = m.hsc_rowptr[p + 1];
            for (int k = r0; k < r1; k++)
                pos[m.hsc_colind[k]] = k;
            for (int i = m.h_pose_ptr[p]; i < m.h_pose_ptr[p + 1]; i++)
            {
                const int a = m.h_pose_edge[i];
                if (!free_free(a))
                    continue;
                const int e1 = m.h_lm_ptr[m.h_e_lm[a] + 1];
                for (int b = a + 1; b < e1; b++)
                    if (free_free(b))
                        off_cnt[pos[m.h_e_pose[b]] + 1]++;
            }
        }
    });
    for (int k = 0; k < B; k++)
        off_cnt[k + 1] += off_cnt[k];
    const size_t Moff = (size_t)off_cnt[B];
    std::vector<int32_t> off_ei(Moff), off_ej(Moff);
    if (!m.splan_on)
    {
        std::vector<int32_t> fill(off_cnt.begin(), off_cnt.end() - 1);
        parallel_rows([&](unsigned, int p0, int p1) {
            std::vector<int32_t> pos(P, -1);
            for (int p = p0; p < p1; p++)
            {
                const int r0 = m.hsc_rowptr[p], r1 = m.hsc_rowptr[p + 1];
                for (int k = r0; k < r1; k++)
                    pos[m.hsc_colind[k]] = k;
                for (int i = m.h_pose_ptr[p]; i < m.h_pose_ptr[p + 1]; i++)
                {
                    const int a = m.h_pose_edge[i];
                    if (!free_free(a))
                        continue;
                    const int e1 = m.h_lm_ptr[m.h_e_lm[a] + 1];
                    for (int b = a + 1; b < e1; b++)
                        if (free_free(b))
                        {
                            const int q = fill[pos[m.h_e_pose[b]]]++;
                            off_ei[q] = a;
                            off_ej[q] = b;
                        }
                }
            }
        });
    }
    laps.lap("structure: product lists");
    // off-diagonal products: from the lists, or (plan) all products minus one per free-free edge
    double n_offdiag = (double)Moff;
    if (m.splan_on)
    {
        double nff = 0;
        for (int e = 0; e < m.E; e++)
            nff += free_free(e);
        n_offdiag = (double)sp.prod.size() - nff;
    }
    if (m.plan_only)
    {
        m.hs = cugo_hsc_struct{};
        m.hs.n_blocks = B;
        prof_[PROF_BUILD_STRUCTURE] += ms_since(t0);
        const auto t1p = Clock::now();
        m.chol.analyze(P, m.hsc_rowptr.data(), m.hsc_colind.data());
        prof_[PROF_SYMBOLIC] += ms_since(t1p);
        fill_structure_stats(B, products, n_offdiag);
        sstats_.schur_slots = m.splan_on ? sp.n_slots : 0;
        return;
    }
    m.d_hsc_rowptr.upload(m.hsc_rowptr, s), m.d_hsc_colind.upload(m.hsc_colind, s);
    m.d_off_ptr.upload(off_cnt, s), m.d_off_ei.upload(off_ei, s), m.d_off_ej.upload(off_ej, s);
    m.d_sys.resize(36 * (size_t)B + 6 * (size_t)P + 16);
    CUGO_HIP(hipStreamSynchronize(s));
    m.hs.n_blocks = B;
    m.hs.d_rowptr = m.d_hsc_rowptr.data(), m.hs.d_colind = m.d_hsc_colind.data();
    m.hs.d_off_ptr = m.d_off_ptr.data(), m.hs.d_off_ei = m.d_off_ei.data();
    m.hs.d_off_ej = m.d_off_ej.data();
    if (m.splan_on)
    {
        m.splan.upload(sp, s);
        m.splan.fill(m.hs);
    }
    else
        SchurPlanDevice::clear(m.hs);
    sstats_.schur_slots = m.splan_on ? sp.n_slots : 0;
    laps.lap("structure: uploads");
    prof_[PROF_BUILD_STRUCTURE] += ms_since(t0);

    const auto t1 = Clock::now();
    m.chol.analyze(P, m.hsc_rowptr.data(), m.hsc_colind.data());
    prof_[PROF_SYMBOLIC] += ms_since(t1);
    laps.lap("structure: symbolic + plan upload");

    fill_structure_stats(B, products, n_offdiag);
    m.structure_dirty = false;
}

void Engine::optimize(int niterations, std::vector<IterRecord>& records, bool verbose)
{
    Impl& m = *impl_;
    if (m.plan_only)
        throw std::runtime_error("cugo: no HIP device in use (plan-only optimiser)");
    hipStream_t s = m.ctx.stream;
    const int maxq = 10;
    const double tau = 1e-5;
    double nu = 2.0, lambda = 0.0, F = 0.0;

    if (m.structure_dirty)
        build_structure();
    const bool sharded = m.world > 1 || m.comm;
    int32_t* d_fail = reinterpret_cast<int32_t*>(m.d_scal.data() + 4);
    auto sync_prof = [&](int item, Clock::time_point t0) {
        if (m.profile)
        {
            CUGO_HIP(hipStreamSynchronize(s));
            prof_[item] += ms_since(t0);
        }
    };

    // Speculative build (single process, CUGO_SPECULATE=0 turns it off): behind the error pass of a first
    // trial the host queues the NEXT iteration's build pass at the trial's estimates before it waits for the
    // trial's result — the 15-20 us the host needs to read F-hat, decide and queue again then overlap with
    // the build instead of leaving the device idle.  The damping that build prepares T for is lambda / 3:
    // what an accepted trial with rho near 1 gives (ref: cuda_graph_optimisation.cpp:97-99, the lower clamp).
    // Accepted with another lambda: the Schur complement recomputes T (as without the fusion).  Rejected:
    // H is rebuilt from the kept estimates before the retry.  Same arithmetic on the same data in every case.
    const bool speculate = !sharded && !m.profile && m.opt.speculate;
    // diagnosis: CUGO_DEBUG_HASH=<file> — position-weighted integer checksums of the arrays every stage of the
    // first trial of an iteration leaves behind, computed by kernels queued in the same stream (no host
    // synchronisation: the flow of the loop stays what it is), appended to the file when optimize() returns
    // (hooks build only, make HOOKS=1: the product build has neither the switch nor the checksum kernel)
#ifdef CUGO_DEBUG_HOOKS
    const char* hash_file = std::getenv("CUGO_DEBUG_HASH");
    if (hash_file)
    {
        m.d_hash.resize(64 * (size_t)std::max(niterations, 1));
        m.d_hash.zero(s);
    }
    auto hash = [&](int iteration, int slot, const void* p, size_t words) {
        if (hash_file)
            cugo_k::launch_hash_words(s, p, words, m.d_hash.data() + 64 * (size_t)iteration + slot);
    };
#else
    auto hash = [](int, int, const void*, size_t) {};
#endif
    const bool trial_event = m.opt.trial_event;                // (CUGO_TRIAL_EVENT=0: wait for the whole stream, A/B)
    const bool trial_poll = !m.profile && m.opt.trial_poll;    // (CUGO_TRIAL_POLL=0: wait by event / stream synchronisation)
    bool have_build = false;      // the build pass of this iteration is already queued
    double built_lambda = -1.0;   // ... with invHll / T for this damping (< 0: none)
    // ... and so is its Schur complement for this damping (< 0: not): the trial that ended the previous iteration took
    // its chi2 from that build pass instead of an error pass of its own (Options::trial_from_build), and the Schur
    // complement for the predicted damping was queued behind the reductions for the host to decide meanwhile
    double schur_ready_lambda = -1.0;
    // Only behind a trial whose predecessor's damping update hit the lower clamp (rho near 1: lambda / 3, the value the
    // speculation predicts): right after a step with a smaller gain ratio the prediction usually misses, and a missed
    // Schur complement costs more than the error pass saves (10k graph: 0.8 ms against 0.05).
    bool prev_clamped = false;

    for (int iteration = 0; iteration < niterations; iteration++)
    {
        const auto it0 = Clock::now();
        // computeErrors + buildSystem fused: chi2 at the current estimates comes out of the
        // build pass (ref: cuda_graph_optimisation.cpp:64-67)
        auto tb = Clock::now();
        // T = Hpl invHll is only materialised for the gather kernels (the landmark-major plan keeps it
        // in LDS).  From the second iteration on the damping of the first trial is known when the build
        // is queued, and the build pass leaves invHll and T for it: that trial's Schur complement then
        // does not read the Hpl stream a second time (CUGO_FUSE_T=0: always the separate edge kernel).
        const bool use_rows = m.rows_on && cugo_k::schur_rows_usable(m.hs, m.max_row_nnz);
        if (!m.splan_on && !use_rows && m.d_T.size() == 0)
            m.d_T.resize((m.ev.block_f32 ? 9 : 18) * (size_t)m.E + 16);
        const bool fuse_allowed = m.opt.fuse_t;
        const bool can_fuse = fuse_allowed && !m.splan_on && m.lm_in_one_group;
        const bool fused_T = have_build ? (can_fuse && built_lambda == lambda) : (can_fuse && iteration > 0);
        const bool ps_on = m.opt.pose_schur && can_fuse && !use_rows && !m.strip_on;
        if (ps_on && m.d_lmrec.size() < 16 * (size_t)std::max(m.L, 1))
            m.d_lmrec.resize(16 * (size_t)std::max(m.L, 1));
        if (!have_build)
        m.timed("build", [&] {
            // chi2 of the build pass is only consumed in the first iteration (see below)
            cugo_k::launch_build(s, m.ev, m.d_poses[m.cur].data(), m.d_lms[m.cur].data(), m.rk,
                                 m.d_Hpp.data(), m.bp(), m.d_Hll.data(), m.bl(), m.d_Hpl.data(),
                                 m.rs(), iteration == 0 ? m.d_scal.data() : nullptr, fused_T ? lambda : -1.0,
                                 fused_T ? m.d_invHll.data() : nullptr,
                                 fused_T && !use_rows ? m.d_T.data() : nullptr, ps_on ? m.d_lmrec.data() : nullptr, ps_on);
            m.hpp_valid = !(ps_on && fused_T);
        });
        have_build = false, built_lambda = -1.0;
        sync_prof(PROF_BUILD_SYSTEM, tb);
        if (iteration == 0)
        {
            const double* hpp = m.d_Hpp.data();
            if (sharded)
            { // diag(Hpp) must be the global sum before taking the maximum
                CUGO_HIP(hipMemcpyAsync(m.d_tmp.data(), m.d_Hpp.data(), 36 * (size_t)m.P * sizeof(double),
                                        hipMemcpyDeviceToDevice, s));
                m.exchange(m.d_tmp.data(), 36 * (size_t)m.P, 0);
                hpp = m.d_tmp.data();
            }
            cugo_k::launch_max_diagonal(s, hpp, m.P, m.d_Hll.data(), m.L, m.rs(),
                                        m.d_scal.data() + 1);
        }
        // chi2 at the current estimates is only read back in the first iteration: afterwards it is
        // the F-hat of the accepted trial (same edges, same estimates), so the host does not have
        // to wait for the build pass before queueing the Schur complement
        if (iteration == 0)
        {
            if (sharded)
            {
                m.exchange(m.d_scal.data(), 1, 0);
                m.exchange(m.d_scal.data() + 1, 1, 1);
            }
            CUGO_HIP(hipMemcpyAsync(m.h_scal.data(), m.d_scal.data(), 2 * sizeof(double),
                                    hipMemcpyDeviceToHost, s));
            CUGO_HIP(hipStreamSynchronize(s));
            F = m.h_scal[0];
            lambda = tau * m.h_scal[1];
        }

        int q = 0;
        double rho = -1.0;
        bool spec_queued = false; // the next iteration's build went out behind this iteration's first trial
        for (; q < maxq && rho < 0; q++)
        {
            if (spec_queued)
            { // that trial was rejected: the speculative pass overwrote H, rebuild it from the kept estimates
                m.timed("build", [&] {
                    cugo_k::launch_build(s, m.ev, m.d_poses[m.cur].data(), m.d_lms[m.cur].data(), m.rk,
                                         m.d_Hpp.data(), m.bp(), m.d_Hll.data(), m.bl(), m.d_Hpl.data(), m.rs(),
                                         nullptr, -1.0, nullptr, nullptr);
                    m.hpp_valid = true;
                });
                spec_queued = false;
            }
            const bool trial_fused = fused_T && q == 0; // invHll and T of this lambda came with the build pass
            if (!(trial_fused && ps_on) && !m.hpp_valid)
            { // the build pass at hand was a fused one (one block stream, no Hpp / Hpl / invHll) and this trial is
              // not the one it was for — the predicted damping missed, or the fused trial was rejected with no
              // speculative build behind it: the two-stream build pass at the estimates in force
                m.timed("build", [&] {
                    cugo_k::launch_build(s, m.ev, m.d_poses[m.cur].data(), m.d_lms[m.cur].data(), m.rk,
                                         m.d_Hpp.data(), m.bp(), m.d_Hll.data(), m.bl(), m.d_Hpl.data(), m.rs(),
                                         nullptr, -1.0, nullptr, nullptr);
                    m.hpp_valid = true;
                });
            }
            const size_t blkw = (m.ev.block_f32 ? 9 : 18) * (size_t)m.E;
            if (q == 0)
            { // what the build pass left
                hash(iteration, 0, m.d_Hpp.data(), 36 * (size_t)m.P);
                hash(iteration, 1, m.d_b.data(), 6 * (size_t)m.P + 3 * (size_t)m.L);
                hash(iteration, 2, m.d_Hll.data(), 9 * (size_t)m.L);
                hash(iteration, 3, m.d_Hpl.data(), blkw);
            }
            auto ts = Clock::now();
            const bool schur_queued = q == 0 && trial_fused && schur_ready_lambda == lambda;
            schur_ready_lambda = -1.0;
            if (!schur_queued)
            m.timed("schur", [&] {
                cugo_k::launch_schur(s, m.ev, m.hs, lambda, 0, m.d_Hpp.data(), m.bp(),
                                     m.d_Hll.data(), m.bl(), m.d_Hpl.data(), m.d_invHll.data(),
                                     m.splan_on || use_rows ? nullptr : m.d_T.data(), m.bsc(), m.Hsc(), trial_fused,
                                     use_rows ? cugo_k::SchurRows{m.d_pose_rec.data(), m.max_row_nnz, nullptr, m.opt.hsc_mfma, m.opt.hsc_xcd}
                                              : cugo_k::SchurRows{nullptr, 0, m.strip_on ? m.d_off_pi.data() : nullptr,
                                                                  m.opt.hsc_mfma, m.opt.hsc_xcd,
                                                                  trial_fused && ps_on ? m.d_lmrec.data() : nullptr,
                                                                  m.d_poses[m.cur].data(), m.rs(), m.bp()});
            });
            if (sharded)
                m.exchange_system();
            sync_prof(PROF_SCHUR, ts);
            if (q == 0)
            {
                hash(iteration, 4, m.d_sys.data(), 36 * (size_t)m.hs.n_blocks + 6 * (size_t)m.P);
                if (m.d_T.size())
                    hash(iteration, 5, m.d_T.data(), blkw);
                hash(iteration, 6, m.d_invHll.data(), 9 * (size_t)m.L);
            }
            auto tn = Clock::now();
#ifdef CUGO_DEBUG_HOOKS
            m.chol.dbg_hash = hash_file && q == 0 ? m.d_hash.data() + 64 * (size_t)iteration : nullptr;
#endif
            m.timed("cholesky", [&] {
                m.chol.factor_solve(m.Hsc(), lambda, m.bsc(), m.xp(), d_fail);
            });
#ifdef CUGO_DEBUG_HOOKS
            m.chol.dbg_hash = nullptr;
#endif
            sync_prof(PROF_NUMERIC, tn);
            if (q == 0)
                hash(iteration, 7, m.xp(), 6 * (size_t)m.P);
            auto tu = Clock::now();
            const int nxt = m.cur ^ 1;
            // single process: the two reductions that end a trial (scale of the update pass, chi2 of
            // the error pass) share one launch, which also drops F-hat, scale and the factorisation flag
            // (slot 4) into the pinned block — no copy is queued behind it.  A shard has to all-reduce
            // the two sums first and keeps the separate launches and the read-back.
            int n_scale_part = 0;
            m.timed("backsubst_update", [&] {
                const bool gform = trial_fused && ps_on; // (the build pass wrote G into d_T and the landmarks' lines)
                n_scale_part = cugo_k::launch_backsubst_update(
                    s, m.ev, lambda, (m.rank == 0 ? lambda : 0.0), m.d_invHll.data(), m.bl(), m.bp(),
                    gform ? m.d_T.data() : m.d_Hpl.data(), m.xp(), m.xl(), m.d_poses[m.cur].data(), m.d_lms[m.cur].data(),
                    m.d_poses[nxt].data(), m.d_lms[nxt].data(), m.rs(), sharded ? m.d_scal.data() + 3 : nullptr,
                    gform ? m.d_lmrec.data() : nullptr);
            });
            sync_prof(PROF_UPDATE, tu);
            if (q == 0)
            {
                hash(iteration, 8, m.xl(), 3 * (size_t)m.L);
                hash(iteration, 9, m.d_poses[nxt].data(), 7 * (size_t)m.Pall);
                hash(iteration, 10, m.d_lms[nxt].data(), 3 * (size_t)m.Lall);
            }
            auto te = Clock::now();
            const double lambda_pred = lambda * (1.0 / 3.0);
            // the trial's chi2 out of the NEXT iteration's build pass: that pass is queued behind a first trial anyway
            // (speculation) and computes the residuals of every edge at the trial's estimates — the error pass would
            // compute them a launch earlier and throw them away
            const bool from_build = m.opt.trial_from_build && speculate && !sharded && trial_poll && q == 0 &&
                                    iteration + 1 < niterations && prev_clamped;
            if (from_build)
            {
                m.timed("build", [&] {
                    cugo_k::launch_build(s, m.ev, m.d_poses[nxt].data(), m.d_lms[nxt].data(), m.rk, m.d_Hpp.data(),
                                         m.bp(), m.d_Hll.data(), m.bl(), m.d_Hpl.data(), m.rs(), nullptr,
                                         can_fuse ? lambda_pred : -1.0, can_fuse ? m.d_invHll.data() : nullptr,
                                         can_fuse && !use_rows ? m.d_T.data() : nullptr, ps_on ? m.d_lmrec.data() : nullptr, ps_on,
                                         true);
                    m.hpp_valid = !(ps_on && can_fuse);
                });
                m.timed("errors", [&] {
                    cugo_k::launch_trial_tail_from_build(s, m.ev, m.rs(), n_scale_part, m.d_scal.data() + 2,
                                                         m.d_scal.data() + 4, m.h_scal.data() + 2, (double)++m.trial_seq,
                                                         reinterpret_cast<unsigned*>(m.d_fail.data() + 2));
                });
                if (can_fuse && !m.splan_on)
                { // the Schur complement of the next iteration's first trial, should the damping be the predicted one:
                  // the host decides while it runs
                    m.timed("schur", [&] {
                        cugo_k::launch_schur(s, m.ev, m.hs, lambda_pred, 0, m.d_Hpp.data(), m.bp(), m.d_Hll.data(), m.bl(),
                                             m.d_Hpl.data(), m.d_invHll.data(), use_rows ? nullptr : m.d_T.data(), m.bsc(),
                                             m.Hsc(), true,
                                             use_rows ? cugo_k::SchurRows{m.d_pose_rec.data(), m.max_row_nnz, nullptr, m.opt.hsc_mfma, m.opt.hsc_xcd}
                                                      : cugo_k::SchurRows{nullptr, 0, m.strip_on ? m.d_off_pi.data() : nullptr,
                                                                          m.opt.hsc_mfma, m.opt.hsc_xcd,
                                                                          ps_on ? m.d_lmrec.data() : nullptr,
                                                                          m.d_poses[nxt].data(), m.rs(), m.bp()});
                    });
                    schur_ready_lambda = lambda_pred;
                }
                spec_queued = true;
            }
            else
            m.timed("errors", [&] {
                if (sharded)
                    cugo_k::launch_errors(s, m.ev, m.d_poses[nxt].data(), m.d_lms[nxt].data(), m.rk,
                                          m.rs(), m.d_scal.data() + 2);
                else
                {
                    cugo_k::launch_errors_tail(s, m.ev, m.d_poses[nxt].data(), m.d_lms[nxt].data(), m.rk, m.rs(),
                                               n_scale_part, m.d_scal.data() + 2, m.d_scal.data() + 4,
                                               m.h_scal.data() + 2, (double)++m.trial_seq,
                                               reinterpret_cast<unsigned*>(m.d_fail.data() + 2));
                }
            });
            m.last_err_buf = nxt;
            sync_prof(PROF_COMPUTE_ERROR, te);
            if (!from_build && speculate && q == 0 && iteration + 1 < niterations)
            {
                if (!trial_poll && trial_event)
                {
                    if (!m.trial_ev)
                        CUGO_HIP(hipEventCreateWithFlags(&m.trial_ev, hipEventDisableTiming));
                    CUGO_HIP(hipEventRecord(m.trial_ev, s));
                }
                m.timed("build", [&] {
                    cugo_k::launch_build(s, m.ev, m.d_poses[nxt].data(), m.d_lms[nxt].data(), m.rk, m.d_Hpp.data(),
                                         m.bp(), m.d_Hll.data(), m.bl(), m.d_Hpl.data(), m.rs(), nullptr,
                                         can_fuse ? lambda_pred : -1.0, can_fuse ? m.d_invHll.data() : nullptr,
                                         can_fuse && !use_rows ? m.d_T.data() : nullptr, ps_on ? m.d_lmrec.data() : nullptr, ps_on);
                    m.hpp_valid = !(ps_on && can_fuse);
                });
                spec_queued = true;
            }
            const bool flag_summed = sharded && m.chol.own_subtrees();
            if (sharded)
            {
                // ranks that factor different subtrees see different zero-pivot flags: as a double the flag
                // rides in the same sum all-reduce as F-hat and the scale
                if (flag_summed)
                    cugo_k::launch_flag_to_double(s, d_fail);
                m.exchange(m.d_scal.data() + 2, flag_summed ? 3 : 2, 0);
                CUGO_HIP(hipMemcpyAsync(m.h_scal.data() + 2, m.d_scal.data() + 2, 3 * sizeof(double),
                                        hipMemcpyDeviceToHost, s));
            }
            // behind a speculative build the host waits for the trial's own last launch only: it then
            // decides and queues the next Schur complement while the build pass still runs
            if (!sharded && trial_poll)
            {
                // the trial's last launch ends by writing the trial's sequence number behind its three words in
                // the pinned block (k_sum_partials2, system-scope release): the host polls that word — no event,
                // no stream synchronisation, and whatever is queued behind the trial keeps running
                const volatile double* seq = m.h_scal.data() + 5;
                const auto w0 = Clock::now();
                auto next_query = w0 + std::chrono::milliseconds(10);
                const auto spin_limit = std::max<Clock::duration>(std::chrono::milliseconds(2), 2 * m.last_trial_wait);
                for (unsigned long spin = 1; *seq != (double)m.trial_seq; spin++)
                {
                    cpu_relax();
                    if ((spin & 0x3FF) != 0)
                        continue;
                    // a trial takes 1 ms (kitti_00 shape) to 3.5 ms (10k graph): the host spins through that — on a
                    // loaded host a yield hands the core away for a scheduler quantum, longer than the trial —; a
                    // wait that lasts longer than twice the previous one (at least 2 ms) offers the core to others
                    // between polls, and every ~10 ms the stream is asked whether it is still working
                    const auto now = Clock::now();
                    if (now - w0 > spin_limit)
                        std::this_thread::yield();
                    if (now < next_query)
                        continue;
                    next_query = now + std::chrono::milliseconds(10);
                    const hipError_t qe = hipStreamQuery(s);
                    if (qe == hipErrorNotReady)
                        continue;
                    CUGO_HIP(qe); // a failed stream will never deliver
                    // the stream is idle: everything queued has run, so the word is there (give the write a moment to
                    // become visible to this core before calling it lost)
                    bool there = false;
                    for (int k = 0; k < 100000 && !there; k++)
                    {
                        std::atomic_thread_fence(std::memory_order_acquire);
                        there = *seq == (double)m.trial_seq;
                        cpu_relax();
                    }
                    if (!there)
                        throw std::runtime_error("cugo: the result of an LM trial never arrived");
                }
                m.last_trial_wait = Clock::now() - w0;
                std::atomic_thread_fence(std::memory_order_acquire);
            }
            else
            {
                if (spec_queued && trial_event)
                    CUGO_HIP(hipEventSynchronize(m.trial_ev));
                else
                    CUGO_HIP(hipStreamSynchronize(s));
                if (!sharded)
                { // the three words in the pinned block are this trial's
                    const volatile double* seq = m.h_scal.data() + 5;
                    for (long spin = 0; *seq != (double)m.trial_seq; spin++)
                    {
                        if (spin == 0)
                            sstats_.trial_sync_retries += 1;
                        if (spin > 2000000000L)
                            throw std::runtime_error("cugo: the result of an LM trial never arrived");
                    }
                    std::atomic_thread_fence(std::memory_order_acquire);
                }
            }
            int32_t fail_flag;
            std::memcpy(&fail_flag, m.h_scal.data() + 4, sizeof fail_flag);
            if (flag_summed)
                fail_flag = m.h_scal[4] > 0.5 ? 1 : 0;
            const bool success = fail_flag == 0;
            const double Fhat = m.h_scal[2];
            const double scale = (success ? m.h_scal[3] : 0.0) + 1e-3;
            const double Fdiff = Fhat - F;
            rho = success ? (F - Fhat) / scale : -1.0;
            if (!success)
                std::printf("factorize failed!\n"); // ref: cuda_linear_solver.cpp:48
            if (rho > 0)
            {
                const double a = 1 - std::pow(2 * rho - 1, 3);
                lambda *= std::max(1.0 / 3.0, std::min(a, 2.0 / 3.0));
                prev_clamped = a <= 1.0 / 3.0;
                nu = 2.0;
                F = Fhat;
                m.cur = nxt; // accept: the trial buffer becomes the estimate (no pop needed)
                if (spec_queued)
                    have_build = true, built_lambda = can_fuse ? lambda_pred : -1.0;
                break;
            }
            else
            {
                lambda *= nu;
                nu *= 2.0;
                prev_clamped = false;
                if (!std::isfinite(lambda) || (success && Fdiff < 1e-4))
                    break;
            }
        }
        records.push_back({iteration, F, lambda, rho, q});
        if (verbose)
            std::printf("iteration= %i;   time(ms): %.4f   chi2= %f;   lambda= %f   rho= %f	   "
                        "nedges= %i    levenberg iterations = %i\n",
                        iteration, ms_since(it0), F, lambda, rho, E_global_, q);
        if (q == maxq || rho < 1e-6 || !std::isfinite(lambda))
            break;
    }
#ifdef CUGO_DEBUG_HOOKS
    if (hash_file)
    {
        std::vector<unsigned long long> h(m.d_hash.size());
        CUGO_HIP(hipMemcpyAsync(h.data(), m.d_hash.data(), h.size() * sizeof(h[0]), hipMemcpyDeviceToHost, s));
        CUGO_HIP(hipStreamSynchronize(s));
        if (FILE* fp = std::fopen(hash_file, "a"))
        {
            std::fprintf(fp, "run\n");
            for (size_t it = 0; it < records.size(); it++)
            {
                for (int k = 0; k < 64; k++)
                    std::fprintf(fp, "%016llx ", h[64 * it + k]);
                std::fprintf(fp, "\n");
            }
            std::fclose(fp);
        }
    }
#endif
    m.collect_times();
}

std::vector<int32_t> Engine::reject_outliers()
{
    Impl& m = *impl_;
    std::vector<int32_t> out;
    if (m.slot_threshold.empty() || m.E == 0)
        return out;
    hipStream_t s = m.ctx.stream;
    // chi2 per slot at the estimates of the last error pass (a rejected trial's estimates if the
    // last trial was rejected: the reference reads whatever d_chiValues holds at that point)
    DevBuf<double> d_chi;
    d_chi.resize((size_t)m.E + 16);
    cugo_k::launch_edge_chi(s, m.ev, m.d_poses[m.last_err_buf].data(), m.d_lms[m.last_err_buf].data(), m.rk,
                            d_chi.data());
    std::vector<double> chi(m.E);
    CUGO_HIP(hipMemcpyAsync(chi.data(), d_chi.data(), sizeof(double) * m.E, hipMemcpyDeviceToHost, s));
    CUGO_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> slots;
    for (int i = 0; i < m.E; i++)
        if (m.slot_edge[i] >= 0 && m.slot_threshold[i] > 0.0 && !(m.h_flags[i] & CUGO_EDGE_INACTIVE) &&
            chi[i] > m.slot_threshold[i])
            slots.push_back(i);
    for (int i : slots)
        out.push_back(m.slot_edge[i]);
    if (m.world > 1)
    { // every rank needs the union: one sum all-reduce of a 0/1 mask over the global edge ids
        std::vector<double> mask(m.Etot, 0.0);
        for (int e : out)
            mask[e] = 1.0;
        DevBuf<double> d_mask;
        d_mask.upload(mask, s);
        m.exchange(d_mask.data(), (size_t)m.Etot, 0);
        CUGO_HIP(hipMemcpyAsync(mask.data(), d_mask.data(), sizeof(double) * m.Etot, hipMemcpyDeviceToHost, s));
        CUGO_HIP(hipStreamSynchronize(s));
        out.clear();
        for (int e = 0; e < m.Etot; e++)
            if (mask[e] > 0.5)
                out.push_back(e);
    }
    else
        std::sort(out.begin(), out.end());
    if (!out.empty())
    { // the flagged edges drop out of a further optimize() on the same initialisation, too
        for (int i : slots)
            m.h_flags[i] |= CUGO_EDGE_INACTIVE;
        m.d_flags.upload(m.h_flags, s);
        if (m.rows_on) // the records of the pose-major list carry the flags
            cugo_k::launch_pose_rec(s, m.ev, m.h_pose_ptr[m.Pall], m.d_pose_rec.data());
        CUGO_HIP(hipStreamSynchronize(s));
        m.structure_dirty = true;
        E_global_ -= (int)out.size();
    }
    return out;
}

void Engine::download(std::vector<double>& poses, std::vector<double>& lms)
{
    Impl& m = *impl_;
    hipStream_t s = m.ctx.stream;
    poses.resize(7 * (size_t)m.Pall);
    lms.resize(3 * (size_t)m.Lall);
    if (m.world > 1 && m.L > 0)
    {
        // landmark estimates live on their owner: zero the others and sum over ranks
        std::vector<double> own(3 * (size_t)m.Lall, 0.0), cur(3 * (size_t)m.Lall);
        CUGO_HIP(hipMemcpyAsync(cur.data(), m.d_lms[m.cur].data(), cur.size() * sizeof(double),
                                hipMemcpyDeviceToHost, s));
        CUGO_HIP(hipStreamSynchronize(s));
        // ownership = the shard's landmark range (ranges partition [0, Lall))
        for (int l = m.shard_l0; l < m.shard_l1; l++)
            for (int k = 0; k < 3; k++)
                own[3 * (size_t)l + k] = cur[3 * (size_t)l + k];
        // NOTE: edges of a landmark never straddle shards (cuts are at landmark boundaries)
        CUGO_HIP(hipMemcpyAsync(m.d_lms[m.cur ^ 1].data(), own.data(), own.size() * sizeof(double),
                                hipMemcpyHostToDevice, s));
        m.exchange(m.d_lms[m.cur ^ 1].data(), own.size(), 0);
        CUGO_HIP(hipMemcpyAsync(lms.data(), m.d_lms[m.cur ^ 1].data(), lms.size() * sizeof(double),
                                hipMemcpyDeviceToHost, s));
    }
    else if (!lms.empty())
        CUGO_HIP(hipMemcpyAsync(lms.data(), m.d_lms[m.cur].data(), lms.size() * sizeof(double),
                                hipMemcpyDeviceToHost, s));
    if (!poses.empty())
        CUGO_HIP(hipMemcpyAsync(poses.data(), m.d_poses[m.cur].data(), poses.size() * sizeof(double),
                                hipMemcpyDeviceToHost, s));
    CUGO_HIP(hipStreamSynchronize(s));
    if (m.world > 1)
    { // keep both buffers consistent for a following initialize-less optimize()
        CUGO_HIP(hipMemcpyAsync(m.d_lms[m.cur].data(), lms.data(), lms.size() * sizeof(double),
                                hipMemcpyHostToDevice, s));
        CUGO_HIP(hipMemcpyAsync(m.d_lms[m.cur ^ 1].data(), lms.data(), lms.size() * sizeof(double),
                                hipMemcpyHostToDevice, s));
        CUGO_HIP(hipStreamSynchronize(s));
    }
}

} // namespace cugo_host
