// CudaGraphOptimisationImpl: flatten the pointer graph, run the engine, write estimates back.
// ref: src/cuda_graph_optimisation.cpp:42-183 (class), src/block_solver.cpp:21-137
// (initialize), src/optimisable_graph.hpp:84-154, 474-572 (index / flag / activeness rules).
#include "../../include/cuda_graph_optimisation.h"

#include <cstring>
#include <stdexcept>

#include "engine.h"

namespace cugo
{

using cugo_host::Engine;
using cugo_host::FlatGraph;

CudaGraphOptimisation::~CudaGraphOptimisation() {}

CudaGraphOptimisation::Ptr CudaGraphOptimisation::create()
{
    return std::make_unique<CudaGraphOptimisationImpl>();
}

CudaGraphOptimisationImpl::CudaGraphOptimisationImpl() : engine_(new Engine) {}

CudaGraphOptimisationImpl::CudaGraphOptimisationImpl(GraphOptimisationOptions& opts)
    : options(opts), engine_(new Engine)
{
}

CudaGraphOptimisationImpl::~CudaGraphOptimisationImpl() {}

void CudaGraphOptimisationImpl::setShard(int rank, int world, ExchangeFn fn, void* user)
{
    engine_->set_shard(rank, world, fn, user);
}

int CudaGraphOptimisationImpl::nActiveEdges() const { return engine_->n_active_edges(); }

std::vector<double> CudaGraphOptimisationImpl::structureStats() const
{
    const auto& s = engine_->structure_stats();
    return {s.hsc_blocks,     s.products,      s.nnzL,          s.chol_flops,  s.supernodes,
            s.stages,         s.front_bytes,   s.offdiag_products, s.up_potrf_flops,
            s.up_trsm_flops,  s.up_syrk_flops, s.up_ea_bytes,   s.backward_bytes};
}

void CudaGraphOptimisationImpl::setKernelTiming(bool on) { engine_->set_kernel_timing(on); }

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

    FlatGraph g;
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

    // ---- edges: every edge with at least one free endpoint is active -------------------
    size_t cap = 0;
    for (BaseEdgeSet* es : edgeSets)
        cap += es->nedges();
    g.e_pose.reserve(cap), g.e_lm.reserve(cap), g.e_flags.reserve(cap);
    g.e_meas.reserve(3 * cap), g.e_omega.reserve(cap), g.e_cam.reserve(cap);
    flatEdges_.clear(), flatEdgeSets_.clear();
    flatEdges_.reserve(cap), flatEdgeSets_.reserve(cap);
    bool omega_uniform = true, rk_set[2] = {false, false}, any_threshold = false;
    cugo_robust rk{CUGO_RK_NONE, 1.0, CUGO_RK_NONE, 1.0};
    for (BaseEdgeSet* es : edgeSets)
    {
        const int dim = es->dim();
        if (dim != 2 && dim != 3)
            throw std::runtime_error("cugo: only 2-d (mono) and 3-d (stereo) BA edge sets are supported");
        const uint8_t stereo_bit = dim == 3 ? CUGO_EDGE_STEREO : 0;
        const RobustKernel& k = es->getRobustKernel();
        if (dim == 3)
            rk.type_stereo = rk_code(k.type()), rk.delta_stereo = k.delta(), rk_set[1] = true;
        else
            rk.type = rk_code(k.type()), rk.delta = k.delta(), rk_set[0] = true;
        size_t nactive = 0;
        const double set_threshold = es->getOutlierThreshold();
        any_threshold = any_threshold || set_threshold > 0.0;
        es->setOutlierCount(0);
        const double set_info = es->getInformation();
        const Camera set_cam = es->getCamera();
        for (BaseEdge* e : es->get())
        {
            if (!e->isActive())
                continue;
            BaseVertex* vp = e->getVertex(0);
            BaseVertex* vl = e->getVertex(1);
            const bool fp = vp->isFixed(), fl = vl->isFixed();
            if (fp && fl)
                continue;
            nactive++;
            flatEdges_.push_back(e), flatEdgeSets_.push_back(es);
            g.e_outlier_threshold.push_back(set_threshold);
            g.e_pose.push_back(vp->getIndex());
            g.e_lm.push_back(vl->getIndex());
            g.e_flags.push_back((uint8_t)((fl ? CUGO_EDGE_FIXED_L : 0) | (fp ? CUGO_EDGE_FIXED_P : 0) | stereo_bit));
            const double* mz = static_cast<const double*>(e->getMeasurement());
            g.e_meas.push_back(mz[0]);
            g.e_meas.push_back(mz[1]);
            g.e_meas.push_back(dim == 3 ? mz[2] : 0.0);
            const double w = options.perEdgeInformation ? (double)e->getInformation() : set_info;
            if (!g.e_omega.empty() && w != g.e_omega[0])
                omega_uniform = false;
            g.e_omega.push_back(w);
            const Camera& c = options.perEdgeCamera ? e->getCamera() : set_cam;
            const double cv[5] = {c.fx, c.fy, c.cx, c.cy, c.bf};
            // deduplicate cameras: last-hit fast path, then a short linear scan
            int ci = -1;
            const size_t ncam = g.cams.size() / 5;
            if (!g.e_cam.empty() && std::memcmp(&g.cams[5 * (size_t)g.e_cam.back()], cv, sizeof cv) == 0)
                ci = g.e_cam.back();
            else
                for (size_t k2 = 0; k2 < ncam; k2++)
                    if (std::memcmp(&g.cams[5 * k2], cv, sizeof cv) == 0)
                    {
                        ci = (int)k2;
                        break;
                    }
            if (ci < 0)
            {
                if (ncam >= 65535)
                    throw std::runtime_error("cugo: more than 65535 distinct cameras");
                ci = (int)ncam;
                g.cams.insert(g.cams.end(), cv, cv + 5);
            }
            g.e_cam.push_back((uint16_t)ci);
        }
        es->setActiveEdgeCount(nactive);
        es->setDirtyState(false);
    }
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
    (void)rk_set;

    engine_->initialize(std::move(g));
    stats_.clear();
    trace_.clear();
}

void CudaGraphOptimisationImpl::optimize(int niterations)
{
    std::vector<cugo_host::IterRecord> rec;
    engine_->optimize(niterations, rec, verbose);
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
    std::vector<double> poses, lms;
    engine_->download(poses, lms);
    for (BaseVertexSet* vs : vertexSets)
        vs->scatterEstimates(vs->isMarginilised() ? lms.data() : poses.data());
}

const TimeProfile& CudaGraphOptimisationImpl::timeProfile()
{
    timeProfile_.clear();
    for (int i = 0; i < cugo_host::PROF_COUNT; i++)
        timeProfile_[Engine::profile_name(i)] = engine_->profile_ms()[i];
    return timeProfile_;
}

} // namespace cugo
