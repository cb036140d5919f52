// CudaGraphOptimisationImpl: flatten the pointer graph, run the engine, write estimates back.
// ref: src/cuda_graph_optimisation.cpp:42-183 (class), src/block_solver.cpp:21-137
// (initialize), src/optimisable_graph.hpp:84-154, 474-572 (index / flag / activeness rules).
#include "../../include/cuda_graph_optimisation.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <stdexcept>

#include "engine.h"
#include "options.h"
#include "thread_pool.h"

namespace cugo
{

namespace detail
{
unsigned hostPoolThreads()
{
    return cugo_host::pool_threads();
}
void hostPoolRun(unsigned chunks, void (*fn)(void*, unsigned), void* ctx)
{
    cugo_host::pool_run(chunks, fn, ctx);
}
} // namespace detail

using cugo_host::Engine;
using cugo_host::FlatGraph;

CudaGraphOptimisation::~CudaGraphOptimisation() {}

CudaGraphOptimisation::Ptr CudaGraphOptimisation::create()
{
    return std::make_unique<CudaGraphOptimisationImpl>();
}

CudaGraphOptimisationImpl::CudaGraphOptimisationImpl() : engine_(new Engine) {}

CudaGraphOptimisationImpl::CudaGraphOptimisationImpl(GraphOptimisationOptions& opts)
    : options(opts), engine_(new Engine(opts.planOnly))
{
}

CudaGraphOptimisationImpl::~CudaGraphOptimisationImpl() {}

void CudaGraphOptimisationImpl::setShard(int rank, int world, ExchangeFn fn, void* user)
{
    engine_->set_shard(rank, world, fn, user);
    flattenValid_ = false; // slot layout, landmark range and Hsc lists were built for the old rank / world
}

void CudaGraphOptimisationImpl::setComm(std::shared_ptr<cugo_host::RcclComm> comm)
{
    engine_->set_comm(std::move(comm));
    flattenValid_ = false;
}

void CudaGraphOptimisationImpl::exchangeStats(double& bytes, int& calls) const
{
    engine_->exchange_stats(bytes, calls);
}

int CudaGraphOptimisationImpl::nActiveEdges() const { return engine_->n_active_edges(); }

std::vector<double> CudaGraphOptimisationImpl::structureStats() const
{
    const auto& s = engine_->structure_stats();
    return {s.hsc_blocks,     s.products,      s.nnzL,          s.chol_flops,  s.supernodes,
            s.stages,         s.front_bytes,   s.offdiag_products, s.up_potrf_flops,
            s.up_trsm_flops,  s.up_syrk_flops, s.up_ea_bytes,   s.backward_bytes, s.schur_slots,
            s.chol_rank_flops, s.chol_top_flops, s.chol_bcast_bytes, s.chol_bcasts, s.trial_sync_retries,
            s.xchg_sys_bytes, s.xchg_sys_full_bytes};
}

void CudaGraphOptimisationImpl::setKernelTiming(int mode) { engine_->set_kernel_timing(mode); }

bool CudaGraphOptimisationImpl::setOption(const char* name, int value)
{
    cugo_host::Options& o = engine_->options();
    const std::string n = name ? name : "";
    if (n == "flatten_reuse")
        o.flatten_reuse = value != 0;
    else if (n == "structure_reuse")
        o.structure_reuse = value != 0;
    else if (n == "init_timing")
        o.init_timing = value != 0;
    else
        return false;
    return true;
}

void CudaGraphOptimisationImpl::kernelTimes(std::vector<std::string>& names, std::vector<double>& ms,
                                            std::vector<int>& launches) const
{
    names.clear(), ms.clear(), launches.clear();
    for (const auto& k : engine_->kernel_times())
    {
        names.push_back(k.name);
        ms.push_back(k.ms);
        launches.push_back(k.launches);
    }
}

static int rk_code(RobustKernelType t)
{
    switch (t)
    {
    case RobustKernelType::Cauchy:
        return CUGO_RK_CAUCHY;
    case RobustKernelType::Tukey:
        return CUGO_RK_TUKEY;
    case RobustKernelType::Huber:
        return CUGO_RK_HUBER;
    default:
        return CUGO_RK_NONE;
    }
}

void CudaGraphOptimisationImpl::initialize()
{
    if (vertexSets.empty() || edgeSets.empty())
        throw std::runtime_error("cugo: initialize() needs at least one vertex set and one edge set");

    const cugo_host::Options& eopt = engine_->options();
    const bool timing = eopt.init_timing;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing)
            return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[cugo init] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - lap_t).count());
        lap_t = n;
    };
    engine_->set_float32_blocks(options.useFloat32);
    FlatGraph& g = engine_->staging();
    // ---- unchanged graph: only the estimates are refreshed -------------------------------
    // If no vertex set or edge set has counted a change since the last full flattening (see
    // "change tracking" in optimisable_graph.h: everything except vertex estimates counts), the
    // flattened, landmark-major graph on the device still is the graph held by these objects.
    // This is the reference fork's isDirty idea (src/block_solver.cpp:151-216 skips the structure
    // rebuild for clean edge sets) carried over to the flattening itself: SLAM back ends — and the
    // reference's own sample, main.cpp:168-190 — call initialize() again on an unchanged graph.
    // CUGO_NO_FLATTEN_REUSE=1 turns it off.
    {
        bool same = flattenValid_ && eopt.flatten_reuse && eopt.structure_reuse &&
                    flattenOptions_[0] == options.perEdgeInformation && flattenOptions_[1] == options.perEdgeCamera &&
                    flattenOptions_[2] == options.useFloat32 &&
                    flattenCounts_.size() == vertexSets.size() + edgeSets.size();
        size_t q = 0;
        for (size_t i = 0; same && i < vertexSets.size(); i++, q++)
            same = flattenCounts_[q].first == vertexSets[i] && flattenCounts_[q].second == vertexSets[i]->changeCount();
        for (size_t i = 0; same && i < edgeSets.size(); i++, q++)
            same = flattenCounts_[q].first == edgeSets[i] && flattenCounts_[q].second == edgeSets[i]->changeCount();
        if (same)
        {
            // (straight into the engine's pinned staging: the copy to the device is then a DMA transfer nobody waits
            // for; a plan-only optimiser has no device and no pinned memory)
            double* hp = options.planOnly ? g.poses.data() : engine_->pinned_poses();
            double* hl = options.planOnly ? g.lms.data() : engine_->pinned_lms();
            for (BaseVertexSet* vs : vertexSets)
                vs->gatherEstimates(vs->isMarginilised() ? hl : hp);
            lap("graph: estimates only");
            for (BaseEdgeSet* es : edgeSets)
                es->setOutlierCount(0);
            if (options.planOnly)
                engine_->refresh_estimates(g);
            else
                engine_->refresh_estimates_pinned();
            flattenReuses_++;
            lap("graph: engine refresh");
            stats_.clear();
            trace_.clear();
            return;
        }
        flattenValid_ = false;
    }
    g.cams.clear();
    // ---- vertex indices: free first (ascending id), fixed after -------------------------
    int nPfree = 0, nLfree = 0, nP = 0, nL = 0;
    for (BaseVertexSet* vs : vertexSets)
    {
        vs->clearEstimates();
        (vs->isMarginilised() ? nLfree : nPfree) += vs->countFree();
        (vs->isMarginilised() ? nL : nP) += (int)vs->size();
    }
    {
        int pf = 0, px = nPfree, lf = 0, lx = nLfree;
        for (BaseVertexSet* vs : vertexSets)
        {
            int a = 0, b = 0;
            if (!vs->isMarginilised())
            {
                if (vs->estimateDim() != 7)
                    throw std::runtime_error("cugo: pose vertex sets must hold Se3D estimates");
                vs->assignIndices(pf, px, a, b);
                pf += a, px += b;
            }
            else
            {
                if (vs->estimateDim() != 3)
                    throw std::runtime_error("cugo: landmark vertex sets must hold Vec3d estimates");
                vs->assignIndices(lf, lx, a, b);
                lf += a, lx += b;
            }
        }
    }
    g.Pall = nP, g.Lall = nL, g.P = nPfree, g.L = nLfree;
    g.poses.resize(7 * (size_t)nP);
    g.lms.resize(3 * (size_t)nL);
    for (BaseVertexSet* vs : vertexSets)
        vs->gatherEstimates(vs->isMarginilised() ? g.lms.data() : g.poses.data());

    lap("graph: vertices");
    // ---- edges: every edge with at least one free endpoint is active -------------------
    // Walking 561k edge objects through virtual getters is the bulk of initialize(); the walk of
    // each edge set is split over worker threads.  Every thread writes its contiguous chunk
    // straight into the final arrays (sized for all edges up front, so the first touch of the
    // pages is parallel too); chunks with skipped edges are compacted afterwards, in order, so
    // the flattened order is exactly the container order.
    size_t cap = 0;
    for (BaseEdgeSet* es : edgeSets)
        cap += es->nedges();
    g.e_pose.resize(cap), g.e_lm.resize(cap), g.e_flags.resize(cap);
    g.e_meas.resize(3 * cap), g.e_omega.resize(cap), g.e_cam.resize(cap);
    g.e_outlier_threshold.resize(cap);
    flatEdges_.resize(cap), flatEdgeSets_.resize(cap);
    bool omega_uniform = true, any_threshold = false;
    cugo_robust rk{CUGO_RK_NONE, 1.0, CUGO_RK_NONE, 1.0};
    struct Chunk
    {
        size_t begin = 0, count = 0;  // slot range written: [begin, begin + count)
        std::vector<double> cams;     // distinct cameras of this chunk (5 each), first-seen order
        bool uniform = true;
        bool too_many_cams = false;
    };
    const unsigned hw = cugo_host::pool_threads();
    lap("graph: edge array alloc");
    size_t out = 0; // slots filled so far (compacted)
    for (BaseEdgeSet* es : edgeSets)
    {
        const int dim = es->dim();
        if (dim != 2 && dim != 3)
            throw std::runtime_error("cugo: only 2-d (mono) and 3-d (stereo) BA edge sets are supported");
        const uint8_t stereo_bit = dim == 3 ? CUGO_EDGE_STEREO : 0;
        const RobustKernel& k = es->robustKernelData();
        if (dim == 3)
            rk.type_stereo = rk_code(k.type()), rk.delta_stereo = k.delta();
        else
            rk.type = rk_code(k.type()), rk.delta = k.delta();
        const double set_threshold = es->getOutlierThreshold();
        any_threshold = any_threshold || set_threshold > 0.0;
        es->setOutlierCount(0);
        const double set_info = es->informationValue();
        const Camera set_cam = es->cameraData();
        const EdgeContainer& ec = es->get();
        const size_t n = ec.size();
        const unsigned nthreads = n < 20000 ? 1u : hw;
        std::vector<Chunk> chunks(nthreads);
        const bool perInfo = options.perEdgeInformation, perCam = options.perEdgeCamera;
        const size_t base = out; // this set's edges go to slots [base, base + n) before compaction
        auto work = [&](unsigned t) {
            Chunk& c = chunks[t];
            const size_t i0 = n * t / nthreads, i1 = n * (t + 1) / nthreads;
            c.begin = base + i0;
            size_t o = c.begin;
            auto it = ec.begin() + i0;
            uint16_t last_cam = 0;
            for (size_t i = i0; i < i1; ++i, ++it)
            {
                // the walk is a chain of cache misses (edge object, then its two vertices):
                // fetch the edge 16 ahead and the vertices of the edge 8 ahead
                if (i + 16 < i1)
                    __builtin_prefetch(*(it + 16));
                if (i + 8 < i1)
                {
                    BaseEdge* e8 = *(it + 8);
                    __builtin_prefetch(e8->getVertex(0));
                    __builtin_prefetch(e8->getVertex(1));
                }
                BaseEdge* e = *it;
                if (!e->isActive())
                    continue;
                BaseVertex* vp = e->getVertex(0);
                BaseVertex* vl = e->getVertex(1);
                const bool fp = vp->isFixed(), fl = vl->isFixed();
                if (fp && fl)
                    continue;
                flatEdges_[o] = e, flatEdgeSets_[o] = es;
                g.e_outlier_threshold[o] = set_threshold;
                g.e_pose[o] = vp->getIndex();
                g.e_lm[o] = vl->getIndex();
                g.e_flags[o] = (uint8_t)((fl ? CUGO_EDGE_FIXED_L : 0) | (fp ? CUGO_EDGE_FIXED_P : 0) | stereo_bit);
                const double* mz = static_cast<const double*>(e->measurementData());
                g.e_meas[3 * o] = mz[0];
                g.e_meas[3 * o + 1] = mz[1];
                g.e_meas[3 * o + 2] = dim == 3 ? mz[2] : 0.0;
                const double w = perInfo ? (double)e->informationValue() : set_info;
                if (o > c.begin && w != g.e_omega[c.begin])
                    c.uniform = false;
                g.e_omega[o] = w;
                const Camera& cm = perCam ? e->cameraData() : set_cam;
                const double cv[5] = {cm.fx, cm.fy, cm.cx, cm.cy, cm.bf};
                // deduplicate cameras: last-hit fast path, then a short linear scan
                int ci = -1;
                const size_t ncam = c.cams.size() / 5;
                if (ncam > 0 && std::memcmp(&c.cams[5 * (size_t)last_cam], cv, sizeof cv) == 0)
                    ci = last_cam;
                else
                    for (size_t k2 = 0; k2 < ncam; k2++)
                        if (std::memcmp(&c.cams[5 * k2], cv, sizeof cv) == 0)
                        {
                            ci = (int)k2;
                            break;
                        }
                if (ci < 0)
                {
                    if (ncam >= 65535)
                    {
                        c.too_many_cams = true;
                        ci = 0;
                    }
                    else
                    {
                        ci = (int)ncam;
                        c.cams.insert(c.cams.end(), cv, cv + 5);
                    }
                }
                last_cam = (uint16_t)ci;
                g.e_cam[o] = (uint16_t)ci; // chunk-local index, remapped below
                o++;
            }
            c.count = o - c.begin;
        };
        struct WorkCtx
        {
            decltype(work)* w;
        } wctx{&work};
        cugo_host::pool_run(
            nthreads, [](void* p, unsigned t) { (*static_cast<WorkCtx*>(p)->w)(t); }, &wctx);
        size_t nactive = 0;
        for (Chunk& c : chunks)
        {
            if (c.too_many_cams)
                throw std::runtime_error("cugo: more than 65535 distinct cameras");
            // merge this chunk's cameras into the global table (first-seen order is kept)
            std::vector<uint16_t> remap(c.cams.size() / 5);
            bool identity = true;
            for (size_t q = 0; q < remap.size(); q++)
            {
                int ci = -1;
                const size_t ncam = g.cams.size() / 5;
                for (size_t k2 = 0; k2 < ncam; k2++)
                    if (std::memcmp(&g.cams[5 * k2], &c.cams[5 * q], 5 * sizeof(double)) == 0)
                    {
                        ci = (int)k2;
                        break;
                    }
                if (ci < 0)
                {
                    if (ncam >= 65535)
                        throw std::runtime_error("cugo: more than 65535 distinct cameras");
                    ci = (int)ncam;
                    g.cams.insert(g.cams.end(), c.cams.begin() + 5 * q, c.cams.begin() + 5 * q + 5);
                }
                remap[q] = (uint16_t)ci;
                identity = identity && ci == (int)q;
            }
            if (c.count > 0 && (!c.uniform || (out > 0 && g.e_omega[c.begin] != g.e_omega[0])))
                omega_uniform = false;
            if (!identity)
                for (size_t i = c.begin; i < c.begin + c.count; i++)
                    g.e_cam[i] = remap[g.e_cam[i]];
            if (c.begin != out && c.count > 0)
            { // edges were skipped in an earlier chunk: close the gap
                std::memmove(&g.e_pose[out], &g.e_pose[c.begin], c.count * sizeof(int32_t));
                std::memmove(&g.e_lm[out], &g.e_lm[c.begin], c.count * sizeof(int32_t));
                std::memmove(&g.e_flags[out], &g.e_flags[c.begin], c.count);
                std::memmove(&g.e_meas[3 * out], &g.e_meas[3 * c.begin], 3 * c.count * sizeof(double));
                std::memmove(&g.e_omega[out], &g.e_omega[c.begin], c.count * sizeof(double));
                std::memmove(&g.e_cam[out], &g.e_cam[c.begin], c.count * sizeof(uint16_t));
                std::memmove(&g.e_outlier_threshold[out], &g.e_outlier_threshold[c.begin], c.count * sizeof(double));
                std::memmove(&flatEdges_[out], &flatEdges_[c.begin], c.count * sizeof(BaseEdge*));
                std::memmove(&flatEdgeSets_[out], &flatEdgeSets_[c.begin], c.count * sizeof(BaseEdgeSet*));
            }
            out += c.count;
            nactive += c.count;
        }
        // the next set starts right behind the compacted edges of this one
        es->setActiveEdgeCount(nactive);
        es->setDirtyState(false);
    }
    g.e_pose.resize(out), g.e_lm.resize(out), g.e_flags.resize(out), g.e_meas.resize(3 * out);
    g.e_omega.resize(out), g.e_cam.resize(out), g.e_outlier_threshold.resize(out);
    flatEdges_.resize(out), flatEdgeSets_.resize(out);
    if (omega_uniform && !g.e_omega.empty())
        g.e_omega.resize(1);
    if (!any_threshold)
        g.e_outlier_threshold.clear();
    if (g.cams.empty())
    {
        const double z[5] = {1, 1, 0, 0, 0};
        g.cams.assign(z, z + 5);
    }
    g.rk = rk;
    lap("graph: edge flatten");

    engine_->initialize(g);
    lap("graph: engine initialize");
    stats_.clear();
    trace_.clear();
    // what this flattening was made from (compared by the next initialize())
    flattenCounts_.clear();
    for (BaseVertexSet* vs : vertexSets)
        flattenCounts_.emplace_back(vs, vs->changeCount());
    for (BaseEdgeSet* es : edgeSets)
        flattenCounts_.emplace_back(es, es->changeCount());
    flattenOptions_[0] = options.perEdgeInformation, flattenOptions_[1] = options.perEdgeCamera;
    flattenOptions_[2] = options.useFloat32;
    flattenValid_ = true;
}

void CudaGraphOptimisationImpl::optimize(int niterations)
{
    const bool timing = engine_->options().init_timing;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing)
            return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[cugo optimize] %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - lap_t).count());
        lap_t = n;
    };
    if (options.planOnly)
        throw std::runtime_error("cugo: no HIP device in use: this optimiser is plan-only (GraphOptimisationOptions::planOnly)");
    std::vector<cugo_host::IterRecord> rec;
    engine_->optimize(niterations, rec, verbose);
    // the walk that writes the estimates back into the vertex objects is ~0.1 ms away (behind the download): the
    // workers of the host pool, parked during the optimisation, start waking up now
    cugo_host::pool_prewake();
    lap("engine optimize");
    for (const auto& r : rec)
    {
        stats_.addStat({r.iteration, r.chi2});
        trace_.push_back({r.lambda, r.rho, r.trials});
    }
    if (shouldProfile_)
        for (const auto& kv : timeProfile())
            std::printf("%s:  %f\n", kv.first.c_str(), kv.second);
    // ref: solver_->updateEdges(edgeSets) (cuda_graph_optimisation.cpp:151): edges whose chi2
    // exceeds their set's outlier threshold are inactivated; the set becomes dirty
    for (int32_t id : engine_->reject_outliers())
    {
        flatEdges_[id]->inactivate();
        BaseEdgeSet* es = flatEdgeSets_[id];
        es->setOutlierCount(es->getOutlierCount() + 1);
        es->setDirtyState(true);
    }
    // ref: finalize(): estimates go back into the user's vertex objects
    const double *hp = nullptr, *hl = nullptr;
    if (!engine_->download_pinned(&hp, &hl))
    { // a landmark shard: the estimates are combined over the ranks first
        std::vector<double>&poses = downloadPoses_, &lms = downloadLms_; // (kept between calls: 24 MB of fresh pages
        engine_->download(poses, lms);                                    //  per call on the 1 M-landmark graph otherwise)
        hp = poses.data(), hl = lms.data();
    }
    lap("outliers + download");
    for (BaseVertexSet* vs : vertexSets)
        vs->scatterEstimates(vs->isMarginilised() ? hl : hp);
    lap("scatter estimates");
}

const TimeProfile& CudaGraphOptimisationImpl::timeProfile()
{
    timeProfile_.clear();
    for (int i = 0; i < cugo_host::PROF_COUNT; i++)
        timeProfile_[Engine::profile_name(i)] = engine_->profile_ms()[i];
    return timeProfile_;
}

} // namespace cugo
