// Public optimiser class of the cugo API, MI355X build
// (ref: include/cuda_graph_optimisation.h:52-282 — same class / method names and semantics).
// The reference header leaks cuda_runtime.h through device_buffer.h / cuda_device.h; here
// the GPU runtime is hidden behind an opaque engine inside libcugo_hip.so.
#pragma once
#include <cassert>
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ba_types.h"

namespace cugo_host
{
class Engine;
class RcclComm;
}

namespace cugo
{

template <class T>
using UniquePtr = std::unique_ptr<T>;
using EdgeSetVec = std::vector<BaseEdgeSet*>;
using VertexSetVec = std::vector<BaseVertexSet*>;

/** one record per LM iteration */
struct CUGO_API BatchInfo
{
    int iteration; //!< iteration number
    double chi2;   //!< total chi2 after the iteration
};

class CUGO_API BatchStatistics
{
public:
    BatchInfo getStartStats() const { return stats.at(0); }
    BatchInfo getLastStats() const { return stats.at(stats.size() - 1); }
    BatchInfo getStatEntry(const int idx) const { return stats.at(idx); }
    void addStat(const BatchInfo& b) { stats.push_back(b); }
    const std::vector<BatchInfo>& get() { return stats; }
    void clear() { stats.clear(); }

private:
    std::vector<BatchInfo> stats;
};

using TimeProfile = std::map<std::string, double>;

/** per-iteration Levenberg-Marquardt trace (extension; the reference only printf's these) */
struct CUGO_API LmTrace
{
    double lambda, rho;
    int trials;
};

/** all-reduce hook for landmark-sharded multi-GPU runs (one process per GPU):
 *  must reduce `n` doubles at DEVICE address `d_buf` in place over all ranks
 *  (op 0 = sum, 1 = max) before returning. */
using ExchangeFn = void (*)(void* d_buf, std::size_t n, int op, void* user);

class CudaGraphOptimisationImpl;

class CUGO_API CudaGraphOptimisation
{
public:
    using Ptr = UniquePtr<CudaGraphOptimisationImpl>;
    virtual ~CudaGraphOptimisation();
    static Ptr create();

    virtual void initialize() = 0;
    virtual void optimize(int niterations) = 0;
    virtual BatchStatistics& batchStatistics() = 0;
    virtual const TimeProfile& timeProfile() = 0;
    virtual EdgeSetVec& getEdgeSets() = 0;
    virtual size_t nVertices(const int id) = 0;
    virtual void clearEdgeSets() = 0;
    virtual void clearVertexSets() = 0;
    virtual void setVerbose(bool status) = 0;
    virtual void setProfile(bool status) = 0;
};

/**
 * Levenberg-Marquardt bundle adjustment on one MI355X (or one landmark shard of a multi-GPU
 * run).  The optimiser never owns vertices, edges or sets — the caller frees them
 * (ref: include/cuda_graph_optimisation.h:129-130).  vertexSets: non-marginalised = poses,
 * marginalised = landmarks; edge vertex 0 = pose, 1 = landmark.
 */
class CUGO_API CudaGraphOptimisationImpl : public CudaGraphOptimisation
{
public:
    CudaGraphOptimisationImpl(GraphOptimisationOptions& options);
    CudaGraphOptimisationImpl();
    ~CudaGraphOptimisationImpl();

    template <typename T>
    bool addEdgeSet(T* edgeSet)
    {
        assert(edgeSet != nullptr);
        edgeSets.push_back(edgeSet);
        return true;
    }
    template <typename T>
    bool addVertexSet(T* vertexSet)
    {
        assert(vertexSet != nullptr);
        vertexSets.push_back(vertexSet);
        return true;
    }

    EdgeSetVec& getEdgeSets() override { return edgeSets; }
    void initialize() override;
    void optimize(int niterations) override;
    BatchStatistics& batchStatistics() override { return stats_; }
    const TimeProfile& timeProfile() override;
    size_t nVertices(const int id) override { return vertexSets.at(id)->size(); }
    void clearEdgeSets() override
    {
        edgeSets.clear();
        flattenValid_ = false;
    }
    void clearVertexSets() override
    {
        vertexSets.clear();
        flattenValid_ = false;
    }
    void setVerbose(bool status) override { verbose = status; }
    void setProfile(bool status) override { shouldProfile_ = status; }

    // ---- extensions (not in the reference) ----
    void setShard(int rank, int world, ExchangeFn fn, void* user);
    /** native exchange of a landmark-sharded run: the per-trial all-reduce of [Hsc | bsc] runs as an
     *  RCCL collective on the solver's stream (communicator: cugo_comm_create of include/cugo_hip.h) */
    void setComm(std::shared_ptr<cugo_host::RcclComm> comm);
    void exchangeStats(double& bytes, int& calls) const;
    const std::vector<LmTrace>& lmTrace() const { return trace_; }
    int nActiveEdges() const;
    /** B, M, nnz(L), flops, supernodes, stages, front bytes, off-diagonal products, then per
     *  factorisation: potrf / trsm / syrk flops, extend-add bytes, backward bytes */
    std::vector<double> structureStats() const;
    /** HIP-event timing on the solver's stream (diagnostic).  1: an event pair round every kernel group and every
     *  kernel (per-kernel figures; each pair adds a few microseconds to what it brackets); 2: one event per group
     *  boundary — group times that add up exactly to the device time of optimize(); 0: off */
    void setKernelTiming(int mode);
    void kernelTimes(std::vector<std::string>& names, std::vector<double>& ms, std::vector<int>& launches) const;
    /** GraphOptimisationOptions::useFloat32 after construction; takes effect at the next initialize() */
    void setUseFloat32(bool on) { options.useFloat32 = on; }
    /** initialize() calls that found the graph unchanged and refreshed only the estimates */
    int flattenReuses() const { return flattenReuses_; }
    bool useFloat32() const { return options.useFloat32; }
    /** a run-time switch of this optimiser ("flatten_reuse", "structure_reuse", "init_timing"; the optimiser took a
     *  snapshot of the CUGO_* environment variables when it was created); false: unknown name */
    bool setOption(const char* name, int value);

private:
    bool verbose = false;
    bool shouldProfile_ = false;
    GraphOptimisationOptions options;
    VertexSetVec vertexSets;
    EdgeSetVec edgeSets;
    std::unique_ptr<cugo_host::Engine> engine_;
    BatchStatistics stats_;
    std::vector<LmTrace> trace_;
    TimeProfile timeProfile_;
    // change counts of the sets at the last full flattening (initialize() refreshes only the
    // estimates while they stand)
    std::vector<std::pair<const void*, unsigned long long>> flattenCounts_;
    bool flattenOptions_[3] = {false, false, false};
    bool flattenValid_ = false;
    int flattenReuses_ = 0;
    // edges handed to the engine at initialize(), in that order (outlier rejection maps back)
    std::vector<BaseEdge*> flatEdges_;
    std::vector<BaseEdgeSet*> flatEdgeSets_;
    std::vector<double> downloadPoses_, downloadLms_; // staging of the estimates that optimize() writes back
};

} // namespace cugo
