// extern "C" surface of libcugo_hip.so (declared in include/cugo_hip.h).
#include <cstring>
#include <deque>
#include <memory>
#include <unordered_map>

#include "../../../include/cugo_hip.h"
#include "../../include/cuda_graph_optimisation.h"
#include "../kernels/kernels.h"
#include "chol_solver.h"
#include "engine.h"
#include "hip_util.h"
#include "rccl_comm.h"
#include "schur_plan.h"

using namespace cugo_host;

namespace
{
template <typename F>
int guarded(F&& f)
{
    try
    {
        f();
        return CUGO_OK;
    }
    catch (const HipError& e)
    {
        set_last_error(e.what());
        return e.code == hipErrorNoDevice ? CUGO_ERR_NO_DEVICE : CUGO_ERR_HIP;
    }
    catch (const std::exception& e)
    {
        set_last_error(e.what());
        const char* w = e.what();
        return std::strstr(w, "no HIP device") ? CUGO_ERR_NO_DEVICE : CUGO_ERR_INVALID;
    }
}

cugo_k::ReduceScratch scratch_for(cugo_ctx* ctx, const cugo_edges* ev)
{
    const size_t need = cugo_k::reduce_scratch_doubles(ev ? ev->n_edges : 0, ev ? ev->n_poses_free : 0,
                                                       ev ? ev->n_landmarks_free : 0);
    if (ctx->scratch.size() < need)
    {
        CUGO_HIP(hipStreamSynchronize(ctx->stream));
        ctx->scratch.resize(need);
    }
    return {ctx->scratch.data(), ctx->scratch.size()};
}
} // namespace

extern "C" {

const char* cugo_last_error(void) { return get_last_error(); }

int cugo_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int cugo_ctx_create(int device, cugo_ctx** out)
{
    return guarded([&] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
            throw std::runtime_error("cugo: no HIP device available (there is no CPU fallback)");
        if (device >= 0)
            CUGO_HIP(hipSetDevice(device));
        auto* c = new cugo_ctx;
        CUGO_HIP(hipGetDevice(&c->device));
        CUGO_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        *out = c;
    });
}

void cugo_ctx_destroy(cugo_ctx* ctx)
{
    if (!ctx)
        return;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int cugo_ctx_sync(cugo_ctx* ctx)
{
    return guarded([&] { CUGO_HIP(hipStreamSynchronize(ctx->stream)); });
}
void* cugo_ctx_stream(cugo_ctx* ctx) { return ctx->stream; }

int cugo_malloc(void** p, size_t bytes)
{
    return guarded([&] { CUGO_HIP(hipMalloc(p, bytes ? bytes : 16)); });
}
int cugo_free(void* p)
{
    return guarded([&] { CUGO_HIP(hipFree(p)); });
}
int cugo_memcpy_h2d(cugo_ctx* ctx, void* d, const void* h, size_t bytes)
{
    return guarded([&] {
        CUGO_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
        CUGO_HIP(hipStreamSynchronize(ctx->stream));
    });
}
int cugo_memcpy_d2h(cugo_ctx* ctx, void* h, const void* d, size_t bytes)
{
    return guarded([&] {
        CUGO_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
        CUGO_HIP(hipStreamSynchronize(ctx->stream));
    });
}
int cugo_memset(cugo_ctx* ctx, void* d, int value, size_t bytes)
{
    return guarded([&] { CUGO_HIP(hipMemsetAsync(d, value, bytes, ctx->stream)); });
}

// ---------------------------------------------------------------- kernel level ---------
int cugo_compute_active_errors(cugo_ctx* ctx, const cugo_edges* ev, const double* d_poses,
                               const double* d_lms, cugo_robust rk, double* d_chi)
{
    return guarded([&] {
        cugo_k::launch_errors(ctx->stream, *ev, d_poses, d_lms, rk, scratch_for(ctx, ev), d_chi);
        CUGO_HIP(hipGetLastError());
    });
}

int cugo_construct_quadratic_form(cugo_ctx* ctx, const cugo_edges* ev, const double* d_poses,
                                  const double* d_lms, cugo_robust rk, double* d_Hpp, double* d_bp,
                                  double* d_Hll, double* d_bl, void* d_Hpl, double* d_chi)
{
    return guarded([&] {
        cugo_k::launch_build(ctx->stream, *ev, d_poses, d_lms, rk, d_Hpp, d_bp, d_Hll, d_bl, d_Hpl,
                             scratch_for(ctx, ev), d_chi);
        CUGO_HIP(hipGetLastError());
    });
}

int cugo_max_diagonal(cugo_ctx* ctx, const double* d_Hpp, int nP, const double* d_Hll, int nL,
                      double* d_out)
{
    return guarded([&] {
        cugo_k::launch_max_diagonal(ctx->stream, d_Hpp, nP, d_Hll, nL, scratch_for(ctx, nullptr),
                                    d_out);
        CUGO_HIP(hipGetLastError());
    });
}

int cugo_compute_schur(cugo_ctx* ctx, const cugo_edges* ev, const cugo_hsc_struct* hs, double lambda,
                       int damp_hsc_diag, const double* d_Hpp, const double* d_bp,
                       const double* d_Hll, const double* d_bl, const void* d_Hpl,
                       double* d_invHll, void* d_T, double* d_bsc, double* d_Hsc)
{
    return guarded([&] {
        if (!d_T && !hs->d_grp_ptr)
            throw std::runtime_error("cugo_compute_schur: d_T may only be NULL with a landmark-major plan "
                                     "(cugo_hsc_plan_create); the gather kernels read T from memory");
        cugo_k::SchurRows form;
        form.mfma = ctx->opt.hsc_mfma, form.xcd = ctx->opt.hsc_xcd;
        cugo_k::launch_schur(ctx->stream, *ev, *hs, lambda, damp_hsc_diag, d_Hpp, d_bp, d_Hll, d_bl,
                             d_Hpl, d_invHll, d_T, d_bsc, d_Hsc, false, form);
        CUGO_HIP(hipGetLastError());
    });
}

int cugo_backsubst_update(cugo_ctx* ctx, const cugo_edges* ev, double lambda,
                          const double* d_invHll, const double* d_bl, const double* d_bp,
                          const void* d_Hpl, const double* d_xp, double* d_xl,
                          const double* d_poses_in, const double* d_lms_in, double* d_poses_out,
                          double* d_lms_out, double* d_scale)
{
    return guarded([&] {
        cugo_k::launch_backsubst_update(ctx->stream, *ev, lambda, lambda, d_invHll, d_bl, d_bp, d_Hpl,
                                        d_xp, d_xl, d_poses_in, d_lms_in, d_poses_out, d_lms_out,
                                        scratch_for(ctx, ev), d_scale);
        CUGO_HIP(hipGetLastError());
    });
}

// ---------------------------------------------------------------- sparse LL^T ----------
int cugo_chol_create(cugo_ctx* ctx, cugo_chol** out)
{
    return guarded([&] {
        auto* s = new cugo_chol;
        s->ctx = ctx;
        *out = s;
    });
}
void cugo_chol_destroy(cugo_chol* s)
{
    if (s && s->ctx)
        (void)hipStreamSynchronize(s->ctx->stream);
    delete s;
}
int cugo_chol_analyze(cugo_chol* s, int n, const int32_t* rowptr, const int32_t* colind)
{
    return guarded([&] { s->analyze(n, rowptr, colind); });
}
int cugo_chol_factor_solve(cugo_chol* s, const double* d_Hsc, double lambda, const double* d_bsc,
                           double* d_x, int32_t* d_fail)
{
    return guarded([&] {
        if (!s->analyzed || !s->ctx)
            throw std::runtime_error("cugo_chol_factor_solve needs an analysed solver with a device context");
        s->factor_solve(d_Hsc, lambda, d_bsc, d_x, d_fail);
    });
}
#ifdef CUGO_DEBUG_HOOKS // diagnosis entry points: only libcugo_hip_hooks.so exports them (csrc/host/cugo_debug.h)
int cugo_debug_pin_reference(void)
{
    return guarded([&] { cugo_debug_pin_reference_solver(); });
}
int cugo_debug_dump_call(int which, int call, const char* path)
{
    return guarded([&] {
        cugo_chol* s = cugo_debug_solver(which);
        if (!s)
            throw std::runtime_error("cugo_debug_dump_call: no solver has run with CUGO_DEBUG_KEEP");
        s->dump_slot(call, path);
    });
}
int cugo_debug_plan_array(int which, const char* name, const int32_t** out)
{
    cugo_chol* s = cugo_debug_solver(which);
    if (!s)
        return -1;
    return cugo_chol_plan_array(s, name, out);
}
int cugo_debug_dump(const char* dir, int* n_calls)
{
    return guarded([&] {
        const int n = cugo_debug_dump_last_solver(dir);
        if (n_calls)
            *n_calls = n;
    });
}
#endif
int cugo_chol_stats(const cugo_chol* s, double* nnzL, double* flops, int* n_super, int* n_stages,
                    double* front_bytes)
{
    if (nnzL)
        *nnzL = s->plan.nnzL;
    if (flops)
        *flops = s->plan.flops;
    if (n_super)
        *n_super = s->plan.n_super;
    if (n_stages)
        *n_stages = s->plan.n_stages;
    if (front_bytes)
        *front_bytes = 8.0 * (double)s->plan.front_doubles;
    return CUGO_OK;
}
int cugo_chol_plan_sizes(const cugo_chol* s, int* n, int* n_super, int* n_rows_total)
{
    *n = s->plan.n;
    *n_super = s->plan.n_super;
    *n_rows_total = (int)s->plan.rows.size();
    return CUGO_OK;
}
int cugo_chol_plan_get(const cugo_chol* s, int32_t* perm, int32_t* super_ptr, int32_t* rows_ptr,
                       int32_t* rows, int32_t* parent)
{
    const CholPlan& P = s->plan;
    std::memcpy(perm, P.perm.data(), sizeof(int32_t) * P.perm.size());
    std::memcpy(super_ptr, P.super_ptr.data(), sizeof(int32_t) * P.super_ptr.size());
    std::memcpy(rows_ptr, P.rows_ptr.data(), sizeof(int32_t) * P.rows_ptr.size());
    std::memcpy(rows, P.rows.data(), sizeof(int32_t) * P.rows.size());
    std::memcpy(parent, P.sparent.data(), sizeof(int32_t) * P.sparent.size());
    return CUGO_OK;
}

int cugo_chol_plan_array(cugo_chol* s, const char* name, const int32_t** out)
{
    const CholPlan& P = s->plan;
    const std::string n(name);
    const std::vector<int32_t>* v = nullptr;
#define CUGO_PLAN_FIELD(f) \
    if (n == #f)           \
    v = &P.f
    CUGO_PLAN_FIELD(perm);
    CUGO_PLAN_FIELD(super_ptr);
    CUGO_PLAN_FIELD(rows_ptr);
    CUGO_PLAN_FIELD(rows);
    CUGO_PLAN_FIELD(sparent);
    CUGO_PLAN_FIELD(child_ptr);
    CUGO_PLAN_FIELD(child);
    CUGO_PLAN_FIELD(rel_ptr);
    CUGO_PLAN_FIELD(rel);
    CUGO_PLAN_FIELD(ncb);
    CUGO_PLAN_FIELD(nb);
    CUGO_PLAN_FIELD(col0);
    CUGO_PLAN_FIELD(col_front);
    CUGO_PLAN_FIELD(stage_task_ptr);
    CUGO_PLAN_FIELD(task_ptr);
    CUGO_PLAN_FIELD(task_fronts);
    CUGO_PLAN_FIELD(blk_front);
    CUGO_PLAN_FIELD(blk_row);
    CUGO_PLAN_FIELD(blk_col);
    CUGO_PLAN_FIELD(alias_of);
    CUGO_PLAN_FIELD(asm_map);
    CUGO_PLAN_FIELD(wl);
#undef CUGO_PLAN_FIELD
    if (n == "asm_info")
    { // [first assembly item in wl, items, then per front: offset of its map in asm_map or -1]
        s->asm_info.assign(1, P.asm0);
        s->asm_info.push_back(P.nasm);
        for (int64_t o : P.asm_off)
            s->asm_info.push_back((int32_t)o);
        v = &s->asm_info;
    }
    if (n == "blk_trans")
        v = &s->trans32;
    if (!v)
    {
        set_last_error("cugo_chol_plan_array: unknown array " + n);
        return CUGO_ERR_INVALID;
    }
    *out = v->data();
    return (int)v->size();
}

int cugo_shard_range(int n_landmarks_total, const int32_t* edges_per_landmark, int rank, int world,
                     int* l0, int* l1)
{
    return guarded([&] {
        if (world < 1 || rank < 0 || rank >= world)
            throw std::runtime_error("cugo_shard_range: bad rank/world");
        std::vector<int32_t> cnt(n_landmarks_total + 1, 0);
        for (int l = 0; l < n_landmarks_total; l++)
            cnt[l + 1] = cnt[l] + edges_per_landmark[l];
        shard_range(cnt, rank, world, *l0, *l1);
    });
}

// ---------------------------------------------------------------- graph level ----------
struct cugo_graph
{
    cugo::GraphOptimisationOptions options;
    std::unique_ptr<cugo::CudaGraphOptimisationImpl> opt;
    cugo::PoseVertexSet poses{false};
    cugo::LandmarkVertexSet lms{true};
    cugo::MonoEdgeSet mono;
    cugo::StereoEdgeSet stereo;
    std::deque<cugo::PoseVertex> pose_store;
    std::deque<cugo::LandmarkVertex> lm_store;
    std::deque<cugo::MonoEdge> mono_store;
    std::deque<cugo::StereoEdge> stereo_store;
    bool attached = false;
    void attach()
    {
        if (attached)
            return;
        opt->addVertexSet(&poses);
        opt->addVertexSet(&lms);
        opt->addEdgeSet(&mono);
        opt->addEdgeSet(&stereo);
        attached = true;
    }
};

int cugo_graph_create(int per_edge_information, int per_edge_camera, cugo_graph** out)
{
    return guarded([&] {
        auto g = std::make_unique<cugo_graph>();
        g->options.perEdgeInformation = per_edge_information != 0;
        g->options.perEdgeCamera = per_edge_camera != 0;
        g->opt = std::make_unique<cugo::CudaGraphOptimisationImpl>(g->options);
        *out = g.release();
    });
}
struct cugo_hsc_plan
{
    cugo_host::SchurPlanDevice dev;
};
int cugo_hsc_plan_create(cugo_ctx* ctx, int n_edges, int n_poses_free, const int32_t* h_pose, const int32_t* h_lm,
                         const uint8_t* h_flags, const int32_t* h_rowptr, const int32_t* h_colind,
                         cugo_hsc_struct* hs, cugo_hsc_plan** out)
{
    int rc = guarded([&] {
        if (!ctx || !hs || !out)
            throw std::runtime_error("cugo_hsc_plan_create: null argument");
        *out = nullptr;
        cugo_host::SchurPlanDevice::clear(*hs);
        cugo_host::SchurPlanHost h;
        cugo_host::build_schur_plan(n_edges, n_poses_free, h_pose, h_lm, h_flags, h_rowptr, h_colind, h);
        if (!h.usable)
            throw std::invalid_argument("cugo_hsc_plan_create: a landmark's active edges straddle two 256-slot groups");
        auto p = std::make_unique<cugo_hsc_plan>();
        p->dev.upload(h, ctx->stream);
        p->dev.fill(*hs);
        *out = p.release();
    });
    return rc;
}
void cugo_hsc_plan_destroy(cugo_hsc_plan* plan) { delete plan; }

int cugo_graph_create_plan_only(int per_edge_information, int per_edge_camera, cugo_graph** out)
{
    return guarded([&] {
        auto g = std::make_unique<cugo_graph>();
        g->options.perEdgeInformation = per_edge_information != 0;
        g->options.perEdgeCamera = per_edge_camera != 0;
        g->options.planOnly = true;
        g->opt = std::make_unique<cugo::CudaGraphOptimisationImpl>(g->options);
        *out = g.release();
    });
}
void cugo_graph_destroy(cugo_graph* g) { delete g; }

int cugo_graph_add_poses(cugo_graph* g, int n, const int32_t* ids, const double* qt, const uint8_t* fixed)
{
    return guarded([&] {
        for (int i = 0; i < n; i++)
        {
            g->pose_store.emplace_back(ids[i], cugo::Se3D(qt + 7 * (size_t)i, qt + 7 * (size_t)i + 4),
                                       fixed && fixed[i]);
            g->poses.addVertex(&g->pose_store.back());
        }
    });
}
int cugo_graph_add_landmarks(cugo_graph* g, int n, const int32_t* ids, const double* xyz,
                             const uint8_t* fixed)
{
    return guarded([&] {
        for (int i = 0; i < n; i++)
        {
            g->lm_store.emplace_back(ids[i], cugo::Vec3d(xyz + 3 * (size_t)i), fixed && fixed[i]);
            g->lms.addVertex(&g->lm_store.back());
        }
    });
}
int cugo_graph_add_edges(cugo_graph* g, int dim, int n, const int32_t* pose_ids, const int32_t* lm_ids,
                         const double* meas, const double* info, const double* cam5)
{
    return guarded([&] {
        if (dim != 2 && dim != 3)
            throw std::runtime_error("cugo_graph_add_edges: dim must be 2 or 3");
        for (int i = 0; i < n; i++)
        {
            cugo::PoseVertex* vp = g->poses.getVertex(pose_ids[i]);
            cugo::LandmarkVertex* vl = g->lms.getVertex(lm_ids[i]);
            cugo::Camera cam;
            if (cam5)
                cam = cugo::Camera(cam5[5 * (size_t)i], cam5[5 * (size_t)i + 1], cam5[5 * (size_t)i + 2],
                                   cam5[5 * (size_t)i + 3], cam5[5 * (size_t)i + 4]);
            if (dim == 2)
            {
                g->mono_store.emplace_back();
                cugo::MonoEdge& e = g->mono_store.back();
                e.setVertex(vp, 0), e.setVertex(vl, 1);
                e.setMeasurement(cugo::Vec2d(meas + 2 * (size_t)i));
                e.setInformation(info ? info[i] : 0.0);
                if (cam5)
                    e.setCamera(cam);
                g->mono.addEdge(&e);
            }
            else
            {
                g->stereo_store.emplace_back();
                cugo::StereoEdge& e = g->stereo_store.back();
                e.setVertex(vp, 0), e.setVertex(vl, 1);
                e.setMeasurement(cugo::Vec3d(meas + 3 * (size_t)i));
                e.setInformation(info ? info[i] : 0.0);
                if (cam5)
                    e.setCamera(cam);
                g->stereo.addEdge(&e);
            }
        }
    });
}
int cugo_graph_set_camera(cugo_graph* g, int dim, const double* c)
{
    const cugo::Camera cam(c[0], c[1], c[2], c[3], c[4]);
    if (dim == 3)
        g->stereo.setCamera(cam);
    else
        g->mono.setCamera(cam);
    return CUGO_OK;
}
int cugo_graph_set_information(cugo_graph* g, int dim, double info)
{
    if (dim == 3)
        g->stereo.setInformation(info);
    else
        g->mono.setInformation(info);
    return CUGO_OK;
}
int cugo_graph_set_robust_kernel(cugo_graph* g, int dim, int type, double delta)
{
    const cugo::RobustKernelType t = type == CUGO_RK_CAUCHY  ? cugo::RobustKernelType::Cauchy
                                     : type == CUGO_RK_TUKEY ? cugo::RobustKernelType::Tukey
                                     : type == CUGO_RK_HUBER ? cugo::RobustKernelType::Huber
                                                             : cugo::RobustKernelType::None;
    if (dim == 3)
        g->stereo.setRobustKernel(t, delta);
    else
        g->mono.setRobustKernel(t, delta);
    return CUGO_OK;
}
int cugo_graph_set_outlier_threshold(cugo_graph* g, int dim, double threshold)
{
    if (dim == 3)
        g->stereo.setOutlierThreshold(threshold);
    else
        g->mono.setOutlierThreshold(threshold);
    return CUGO_OK;
}
int cugo_graph_n_outliers(cugo_graph* g, int dim)
{
    return (int)(dim == 3 ? g->stereo.getOutlierCount() : g->mono.getOutlierCount());
}
int cugo_graph_get_edge_active(cugo_graph* g, int dim, int n, uint8_t* active)
{
    return guarded([&] {
        if (dim == 3)
            for (int i = 0; i < n && i < (int)g->stereo_store.size(); i++)
                active[i] = g->stereo_store[i].isActive() ? 1 : 0;
        else
            for (int i = 0; i < n && i < (int)g->mono_store.size(); i++)
                active[i] = g->mono_store[i].isActive() ? 1 : 0;
    });
}
int cugo_graph_set_shard(cugo_graph* g, int rank, int world, cugo_exchange_fn fn, void* user)
{
    return guarded([&] { g->opt->setShard(rank, world, fn, user); });
}
int cugo_comm_unique_id(void* id128)
{
    return guarded([&] {
        if (!id128)
            throw std::runtime_error("cugo_comm_unique_id: null buffer");
        cugo_host::RcclComm::unique_id(id128);
    });
}
struct cugo_comm
{
    std::shared_ptr<cugo_host::RcclComm> c;
};
int cugo_comm_create(const void* id128, int rank, int world, cugo_comm** out)
{
    return guarded([&] {
        if (!id128 || !out || world < 1 || rank < 0 || rank >= world)
            throw std::runtime_error("cugo_comm_create: bad arguments");
        auto* h = new cugo_comm;
        try
        {
            h->c = std::make_shared<cugo_host::RcclComm>(id128, rank, world);
        }
        catch (...)
        {
            delete h;
            throw;
        }
        *out = h;
    });
}
void cugo_comm_destroy(cugo_comm* comm) { delete comm; }
int cugo_graph_set_comm(cugo_graph* g, cugo_comm* comm)
{
    return guarded([&] {
        if (!comm)
            throw std::runtime_error("cugo_graph_set_comm: null communicator");
        g->opt->setComm(comm->c);
    });
}
int cugo_graph_exchange_stats(cugo_graph* g, double* bytes, int32_t* calls)
{
    return guarded([&] {
        int c = 0;
        double b = 0;
        g->opt->exchangeStats(b, c);
        if (bytes)
            *bytes = b;
        if (calls)
            *calls = c;
    });
}
int cugo_set_device(int device)
{
    return guarded([&] { CUGO_HIP(hipSetDevice(device)); });
}
int cugo_graph_initialize(cugo_graph* g)
{
    return guarded([&] {
        g->attach();
        g->opt->initialize();
    });
}
int cugo_graph_flatten_reuses(cugo_graph* g)
{
    int n = -1;
    if (guarded([&] {
            if (!g)
                throw std::runtime_error("cugo_graph_flatten_reuses: null graph");
            n = g->opt->flattenReuses();
        }) != 0)
        return -1;
    return n;
}
int cugo_graph_set_option(cugo_graph* g, const char* name, int value)
{
    return guarded([&] {
        if (!g || !g->opt->setOption(name, value))
            throw std::invalid_argument(std::string("cugo_graph_set_option: unknown option ") + (name ? name : "(null)"));
    });
}
int cugo_graph_optimize(cugo_graph* g, int n_iters)
{
    return guarded([&] { g->opt->optimize(n_iters); });
}
int cugo_graph_n_stats(cugo_graph* g) { return (int)g->opt->batchStatistics().get().size(); }
int cugo_graph_get_stats(cugo_graph* g, int32_t* iteration, double* chi2, int cap)
{
    const auto& st = g->opt->batchStatistics().get();
    int n = 0;
    for (; n < (int)st.size() && n < cap; n++)
    {
        iteration[n] = st[n].iteration;
        chi2[n] = st[n].chi2;
    }
    return n;
}
int cugo_graph_get_trace(cugo_graph* g, double* lambda, double* rho, int32_t* trials, int cap)
{
    const auto& tr = g->opt->lmTrace();
    int n = 0;
    for (; n < (int)tr.size() && n < cap; n++)
    {
        lambda[n] = tr[n].lambda;
        rho[n] = tr[n].rho;
        trials[n] = tr[n].trials;
    }
    return n;
}
int cugo_graph_get_poses(cugo_graph* g, int n, const int32_t* ids, double* qt)
{
    return guarded([&] {
        for (int i = 0; i < n; i++)
        {
            const cugo::Se3D& e = g->poses.getVertex(ids[i])->getEstimate();
            e.copyTo(qt + 7 * (size_t)i, qt + 7 * (size_t)i + 4);
        }
    });
}
int cugo_graph_get_landmarks(cugo_graph* g, int n, const int32_t* ids, double* xyz)
{
    return guarded([&] {
        for (int i = 0; i < n; i++)
            g->lms.getVertex(ids[i])->getEstimate().copyTo(xyz + 3 * (size_t)i);
    });
}
int cugo_graph_set_poses(cugo_graph* g, int n, const int32_t* ids, const double* qt)
{
    return guarded([&] {
        for (int i = 0; i < n; i++)
            g->poses.getVertex(ids[i])->setEstimate(cugo::Se3D(qt + 7 * (size_t)i, qt + 7 * (size_t)i + 4));
    });
}
int cugo_graph_set_landmarks(cugo_graph* g, int n, const int32_t* ids, const double* xyz)
{
    return guarded([&] {
        for (int i = 0; i < n; i++)
            g->lms.getVertex(ids[i])->setEstimate(cugo::Vec3d(xyz + 3 * (size_t)i));
    });
}
int cugo_graph_n_active_edges(cugo_graph* g) { return g->opt->nActiveEdges(); }
int cugo_graph_time_profile(cugo_graph* g, char* names, int buf_len, double* ms, int cap)
{
    const auto& tp = g->opt->timeProfile();
    std::string all;
    int n = 0;
    for (const auto& kv : tp)
    {
        if (n >= cap)
            break;
        all += kv.first + "\n";
        ms[n++] = kv.second;
    }
    if (names && buf_len > 0)
    {
        std::strncpy(names, all.c_str(), buf_len - 1);
        names[buf_len - 1] = 0;
    }
    return n;
}
int cugo_graph_set_verbose(cugo_graph* g, int v)
{
    g->opt->setVerbose(v != 0);
    return CUGO_OK;
}
int cugo_graph_set_float32(cugo_graph* g, int on)
{
    g->opt->setUseFloat32(on != 0);
    return CUGO_OK;
}
int cugo_graph_set_kernel_timing(cugo_graph* g, int on)
{
    g->opt->setKernelTiming(on);
    return CUGO_OK;
}
int cugo_graph_kernel_times(cugo_graph* g, char* names, int buf_len, double* ms, int32_t* launches, int cap)
{
    std::vector<std::string> nm;
    std::vector<double> t;
    std::vector<int> c;
    g->opt->kernelTimes(nm, t, c);
    std::string all;
    int n = 0;
    for (; n < (int)nm.size() && n < cap; n++)
    {
        all += nm[n] + "\n";
        ms[n] = t[n];
        launches[n] = c[n];
    }
    if (names && buf_len > 0)
    {
        std::strncpy(names, all.c_str(), buf_len - 1);
        names[buf_len - 1] = 0;
    }
    return n;
}
int cugo_memcpy_d2d(cugo_ctx* ctx, void* d, const void* s, size_t bytes)
{
    return guarded([&] {
        CUGO_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        CUGO_HIP(hipStreamSynchronize(ctx->stream));
    });
}
int cugo_graph_structure_stats(cugo_graph* g, double* out, int cap)
{
    const auto v = g->opt->structureStats();
    int n = 0;
    for (; n < (int)v.size() && n < cap; n++)
        out[n] = v[n];
    return n;
}

} // extern "C"
