// Engine: device-resident BA solver driven by flat arrays.  It is the counterpart of the
// reference's BlockSolver (ref: src/block_solver.{h,cpp}) plus the LM loop of
// CudaGraphOptimisationImpl::optimize (ref: src/cuda_graph_optimisation.cpp:48-154).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/cugo_hip.h"

namespace cugo_host
{

class RcclComm;
struct Options;

// Result of flattening a graph (ref: VertexSet::generateEstimateData + EdgeSet::init,
// src/optimisable_graph.hpp:84-126,474-572).  Indices: free vertices first.
struct FlatGraph
{
    int Pall = 0, Lall = 0, P = 0, L = 0;
    std::vector<double> poses; // Pall x 7
    std::vector<double> lms;   // Lall x 3
    // active edges in ANY order (the engine sorts them landmark-major)
    std::vector<int32_t> e_pose, e_lm;
    std::vector<double> e_meas; // E x 3 (AoS on input)
    std::vector<double> e_omega; // E or 1
    std::vector<uint8_t> e_flags;
    std::vector<uint16_t> e_cam; // E or empty
    std::vector<double> cams;    // n_cams x 5
    cugo_robust rk{CUGO_RK_NONE, 1.0, CUGO_RK_NONE, 1.0};
    // outlier rejection (ref: EdgeSet::updateEdges, optimisable_graph.hpp:603-640): per edge
    // the chi2 threshold of its edge set, 0 = disabled; empty = disabled for all
    std::vector<double> e_outlier_threshold;
    int n_edges() const { return (int)e_pose.size(); }
};

struct IterRecord
{
    int iteration;
    double chi2, lambda, rho;
    int trials;
};

struct StructureStats
{
    double hsc_blocks = 0, products = 0, nnzL = 0, chol_flops = 0, supernodes = 0, stages = 0,
           front_bytes = 0, offdiag_products = 0, up_potrf_flops = 0, up_trsm_flops = 0,
           up_syrk_flops = 0, up_ea_bytes = 0, backward_bytes = 0, schur_slots = 0;
    // sharded run with rank-owned elimination subtrees: factorisation work of this rank's own subtrees / of the
    // replicated top (same unit as chol_flops' model), bytes this rank's factorisation exchanges per trial
    // (update blocks into the top + solution ranges), number of those broadcasts
    double chol_rank_flops = 0, chol_top_flops = 0, chol_bcast_bytes = 0, chol_bcasts = 0;
    double trial_sync_retries = 0; // waits for an LM trial that returned before its result was there (expected: 0)
    // sharded run: bytes of the Schur system [Hsc | bsc] this rank RECEIVES per LM trial — with rank-owned
    // subtrees only its own segment (reduce-scatter) and the top's part (all-reduce) — and what the full
    // all-reduce of the system would hand to it
    double xchg_sys_bytes = 0, xchg_sys_full_bytes = 0;
};

enum ProfItem
{
    PROF_INITIALIZE = 0,
    PROF_BUILD_STRUCTURE,
    PROF_COMPUTE_ERROR,
    PROF_BUILD_SYSTEM,
    PROF_SCHUR,
    PROF_SYMBOLIC,
    PROF_NUMERIC,
    PROF_UPDATE,
    PROF_COUNT
};

// contiguous landmark range of shard `rank`, balanced by edge count; lm_cnt is the
// exclusive prefix sum of edges per landmark (size Lall+1)
void shard_range(const std::vector<int32_t>& lm_cnt, int rank, int world, int& l0, int& l1);

class Engine
{
public:
    // plan_only: no device is touched; initialize() stops after the host-side flattening and
    // builds the structure (Hsc pattern, product lists, ordering, symbolic factor) right away
    explicit Engine(bool plan_only = false);
    ~Engine();
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    // the run-time switches of this optimiser (options.h): a snapshot of the environment taken when it was
    // created; cugo_graph_set_option changes single ones afterwards
    Options& options();
    void set_shard(int rank, int world, cugo_exchange_fn fn, void* user);
    // native exchange: RCCL communicator over the ranks of the job (shared by all optimisers of the
    // process); all-reduces then run on the solver's stream without a host synchronisation
    void set_comm(std::shared_ptr<RcclComm> comm);
    void exchange_stats(double& bytes, int& calls) const; // since the last initialize()
    // ref: BlockSolver::initialize (block_solver.cpp:21-137)
    // float storage of the Hpl / T block streams (GraphOptimisationOptions::useFloat32); takes
    // effect at the next initialize().  The environment variable CUGO_FLOAT32=1 forces it on.
    void set_float32_blocks(bool on);
    void initialize(FlatGraph& g);
    // same graph as at the last initialize(), new estimates (g.poses / g.lms): the flattened edge
    // arrays, the structure and the symbolic factor on the device stay as they are
    void refresh_estimates(const FlatGraph& g);
    // The same through PINNED host staging owned by the engine (7 doubles per pose / 3 per landmark, by index, as in
    // FlatGraph::poses / lms): the caller gathers the estimates straight into pinned_poses() / pinned_lms() and the
    // copy to the device is a DMA transfer this call does not wait for (from pageable memory the runtime stages the
    // copy synchronously: 0.13 ms of the 0.33 ms an estimates-only initialize() took on the kitti_00 shape).
    double* pinned_poses();
    double* pinned_lms();
    void refresh_estimates_pinned();
    // estimates back to the host, into the same pinned staging (single process; false on a landmark shard, whose
    // landmark estimates are summed over the ranks first: download() does that)
    bool download_pinned(const double** poses, const double** lms);
    // a FlatGraph owned by the engine whose buffers survive between initialize() calls
    FlatGraph& staging();
    // ref: optimize(); appends to records. verbose prints one line per iteration.
    void optimize(int niterations, std::vector<IterRecord>& records, bool verbose);
    // estimates back to host (ref: VertexSet::finalise, optimisable_graph.hpp:137-154)
    void download(std::vector<double>& poses, std::vector<double>& lms);
    // ref: BlockSolver::updateEdges at the end of optimize(): edges whose chi2 in the last error
    // pass exceeds their set's threshold become inactive.  Returns their indices in the order of
    // the FlatGraph passed to initialize() (all ranks return the same list).
    std::vector<int32_t> reject_outliers();
    int n_active_edges() const { return E_global_; }
    const StructureStats& structure_stats() const { return sstats_; }
    const double* profile_ms() const { return prof_; }
    static const char* profile_name(int i);
    // per-kernel-group device times measured with HIP events on the engine's stream
    // (enabled by set_kernel_timing; adds event overhead, so not used inside timed runs)
    // mode 1: event pairs round every kernel group and every kernel (per-kernel figures; each pair adds a few us);
    // mode 2: one event per group boundary (group times that add up exactly to the device time of optimize()); 0: off
    void set_kernel_timing(int mode);
    struct KernelTime
    {
        std::string name;
        double ms;
        int launches;
    };
    std::vector<KernelTime> kernel_times() const;
    // device views for kernel-level tests / the C ABI
    struct Impl;
    Impl* impl() { return impl_; }

private:
    void build_structure();
    void fill_structure_stats(int B, double products, double offdiag_products);
    Impl* impl_;
    int E_global_ = 0;
    StructureStats sstats_;
    double prof_[PROF_COUNT] = {0};
};

} // namespace cugo_host
