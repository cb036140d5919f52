/*
 * cugo_hip.h — C ABI of libcugo_hip.so, the MI355X (gfx950) implementation of the bundle-
 * adjustment hot path of KudanLimited/cuda-bundle-adjustment.
 *
 * Two layers, both plain C (pointers + sizes, no C++ or torch types):
 *
 *  (1) kernel-level entry points  cugo_*  — one per free function of the reference's
 *      `namespace cugo::gpu` (src/cuda/cuda_block_solver.h:55-256) and per method of its
 *      linear-solver object (src/cuda_linear_solver.h:31-53).  All pointers named d_* are
 *      DEVICE pointers; work is enqueued on the context's HIP stream and is asynchronous
 *      unless the function returns a host value.
 *  (2) graph-level entry points   cugo_graph_*  — what a foreign-language binding of the
 *      reference's public class (include/cuda_graph_optimisation.h:132-252) would call:
 *      flat host arrays in, BatchStatistics / estimates out.
 *
 * Citations "ref:" are file:line in the reference repository.
 * Every function returns CUGO_OK (0) or a negative error code; cugo_last_error() gives text.
 * There is NO CPU fallback: without a HIP device every call fails with CUGO_ERR_NO_DEVICE.
 */
#ifndef CUGO_HIP_H
#define CUGO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUGO_OK 0
#define CUGO_ERR_NO_DEVICE (-1)
#define CUGO_ERR_HIP (-2)
#define CUGO_ERR_INVALID (-3)
#define CUGO_ERR_NUMERIC (-4) /* zero pivot: ref src/cuda_linear_solver.cpp:44-52 */

/* ref: src/robust_kernel.h:12-17 */
enum { CUGO_RK_NONE = 0, CUGO_RK_CAUCHY = 1, CUGO_RK_TUKEY = 2,
       CUGO_RK_HUBER = 3 /* extension: g2o RobustKernelHuber, rho(x) = x | 2*delta*sqrt(x) - delta^2 */ };

/* edge flag bits; bits 0/1 are the reference's EdgeFlag (ref: src/constants.h:28-32) */
enum
{
    CUGO_EDGE_FIXED_L = 1,
    CUGO_EDGE_FIXED_P = 2,
    CUGO_EDGE_STEREO = 4,  /* 3-d measurement (StereoEdge) instead of 2-d (MonoEdge) */
    CUGO_EDGE_INACTIVE = 8 /* outlier / removed: contributes nothing (ref: outliers[e]!=0) */
};

typedef struct cugo_ctx cugo_ctx; /* device + stream + scratch; replaces CudaDevice */

const char* cugo_last_error(void);
int cugo_device_count(void);
/* ref: src/cuda_device.cpp:166-282 (device pick + streams). device<0 = current device. */
int cugo_ctx_create(int device, cugo_ctx** out);
void cugo_ctx_destroy(cugo_ctx* ctx);
int cugo_ctx_sync(cugo_ctx* ctx);
void* cugo_ctx_stream(cugo_ctx* ctx); /* hipStream_t */

/* ref: src/device_buffer.h:33-276 (RAII device array) — plain allocation helpers */
int cugo_malloc(void** d_ptr, size_t bytes);
int cugo_free(void* d_ptr);
int cugo_memcpy_h2d(cugo_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int cugo_memcpy_d2h(cugo_ctx* ctx, void* h_dst, const void* d_src, size_t bytes); /* syncs */
int cugo_memset(cugo_ctx* ctx, void* d_dst, int value, size_t bytes);

/*
 * Flattened edge data of ALL edge sets of one optimiser, landmark-major:
 * edges are sorted by (landmark index, pose index); the edges of landmark l are
 * [lm_ptr[l], lm_ptr[l+1]).  Mono and stereo edges share the arrays (flag bit STEREO).
 * This replaces the per-set arrays of ref: src/optimisable_graph.hpp:474-601
 * (measurements | edge2PL | flags | omegas | cameras) and edge2Hpl: the Hpl block of edge
 * e is block e (identity map), valid when (flags & 3) == 0.
 * pose_ptr/pose_edge give the same edges grouped by pose (CSR over pose index).
 */
typedef struct cugo_edges
{
    int n_edges;
    int n_poses_total, n_landmarks_total; /* incl. fixed (indices >= n_free are fixed) */
    int n_poses_free, n_landmarks_free;
    const int32_t* d_pose;   /* [E] pose index      (ref edge2PL[e][0]) */
    const int32_t* d_lm;     /* [E] landmark index  (ref edge2PL[e][1]) */
    const double* d_meas;    /* [3][E] planar: u, v, u_right */
    const double* d_omega;   /* [E] or [1] information (ref omegas) */
    int n_omega;             /* E or 1  (ref: perEdgeInformation) */
    const uint8_t* d_flags;  /* [E] CUGO_EDGE_* */
    const uint16_t* d_cam;   /* [E] index into d_cams, or NULL when n_cams == 1 */
    const double* d_cams;    /* [n_cams][5] fx fy cx cy bf (ref cameras, deduplicated) */
    int n_cams;
    const int32_t* d_lm_ptr;    /* [n_landmarks_total+1] */
    const int32_t* d_pose_ptr;  /* [n_poses_total+1]     */
    const int32_t* d_pose_edge; /* [E] edge ids grouped by pose, ascending landmark */
    /* Element type of the two per-edge block arrays d_Hpl and d_T ([E][18] each) handed to
     * cugo_construct_quadratic_form / cugo_compute_schur / cugo_backsubst_update:
     * 0 = double, 1 = float (the fp32-internal mode, ref: the USE_FLOAT32 build option,
     * CMakeLists.txt:8, src/scalar.h:24-28).  Only the storage of these two streams changes:
     * every value is widened on load, all arithmetic and every other array stay fp64. */
    int block_f32;
} cugo_edges;

typedef struct cugo_robust
{
    int type;     /* CUGO_RK_* for 2-d (mono) edges */
    double delta; /* ref: createRkFunction src/cuda/cuda_block_solver.h:61 — passed by value */
    int type_stereo; /* same for 3-d (stereo) edges: per edge set, not process-global */
    double delta_stereo;
} cugo_robust;

/* ---- (1) kernel-level entry points --------------------------------------------------- */

/* ref: gpu::computeActiveErrors_<2|3> (cuda_block_solver.h:179-193; .cu:1060-1133).
 * d_poses: [n_poses_total][7] (qx qy qz qw tx ty tz), d_lms: [n_landmarks_total][3].
 * Writes the total chi2 to d_chi[0] (device).  Deterministic (fixed reduction order). */
int cugo_compute_active_errors(cugo_ctx* ctx, const cugo_edges* ev, const double* d_poses,
                               const double* d_lms, cugo_robust rk, double* d_chi);

/* ref: gpu::constructQuadraticForm_<2|3> (cuda_block_solver.h:159-176; .cu:1152-1220) plus
 * the four fillZero calls of BlockSolver::buildSystem (block_solver.cpp:287-294).
 * Outputs (all column-major blocks, reference layout):
 *   d_Hpp [Pfree][36], d_bp [Pfree][6], d_Hll [Lfree][9], d_bl [Lfree][3], d_Hpl [E][18]
 *   (d_Hpl and, below, d_T: double, or float when ev->block_f32).
 * Also writes chi2 at these estimates to d_chi[0] if d_chi != NULL. No atomics. */
int cugo_construct_quadratic_form(cugo_ctx* ctx, const cugo_edges* ev, const double* d_poses,
                                  const double* d_lms, cugo_robust rk, double* d_Hpp,
                                  double* d_bp, double* d_Hll, double* d_bl, void* d_Hpl,
                                  double* d_chi);

/* ref: gpu::maxDiagonal x2 (cuda_block_solver.h:78-81; block_solver.cpp:309-320).
 * d_out[0] = max(0, max diag(Hpp), max diag(Hll)). */
int cugo_max_diagonal(cugo_ctx* ctx, const double* d_Hpp, int n_poses, const double* d_Hll,
                      int n_landmarks, double* d_out);

/* Hsc block structure on the device (ref: HschurSparseBlockMatrix,
 * src/sparse_block_matrix.cpp:63-156, and mulBlockIds .cu:1347-1378).
 * Upper-triangular block CSR, diagonal block first in each row.  Off-diagonal block k has
 * the contribution list [d_off_ptr[k], d_off_ptr[k+1]) of (edge_i, edge_j) pairs:
 * Hsc[k] -= T[edge_i] * Hpl[edge_j]^T. (diagonal blocks use the pose's own edge list).
 * ZERO-INITIALISE the struct (memset / = {0}) before filling it: cugo_compute_schur selects the
 * landmark-major plan kernels when the plan fields (d_grp_ptr ...) are non-NULL. */
typedef struct cugo_hsc_struct
{
    int n_blocks;             /* B */
    const int32_t* d_rowptr;  /* [Pfree+1] */
    const int32_t* d_colind;  /* [B] */
    const int32_t* d_off_ptr; /* [B+1] (empty range for diagonal blocks) */
    const int32_t* d_off_ei;  /* [M_off] */
    const int32_t* d_off_ej;  /* [M_off] */
    /* Landmark-major product plan (optional: with d_grp_ptr == NULL cugo_compute_schur gathers the
     * products destination-major through the contribution lists above).  A group = 256 consecutive
     * edge slots; the products of a group's landmarks are summed per (group, Hsc block) into a
     * partial slot, the slots of every block are then added in group order.  Built by
     * cugo_hsc_plan_create(); requires that no landmark has active edges in two groups. */
    int n_groups;                /* ceil(n_edges / 256) */
    int n_slots, n_rhs;
    const int32_t* d_grp_ptr;    /* [n_groups+1] partial slots of each group (longest product list first) */
    const int32_t* d_grp_nwave;  /* [n_groups] leading slots of the group that a whole wave works on */
    const int32_t* d_slot_rhs;   /* [n_slots] index of the slot's rhs partial (diagonal block) or -1 */
    const int32_t* d_slot_ptr;   /* [n_slots+1] product range of each slot */
    const uint16_t* d_prod;      /* [M] a | b << 8: slots (within the group) of the T and Hpl operand */
    const int32_t* d_red_ptr;    /* [B+1] partial slots of every Hsc block ... */
    const int32_t* d_red_slot;   /* [n_slots] ... in group order */
    const int32_t* d_blk_pose;   /* [B] pose of a diagonal block, -1 otherwise */
    double* d_part_H;            /* [n_slots][36] scratch */
    double* d_part_b;            /* [n_rhs][6] scratch */
} cugo_hsc_struct;

/* Builds the landmark-major plan of a flattened edge set on the host and uploads it (ref: the device
 * structure set-up findHschureMulBlockIndices, src/cuda/cuda_block_solver.cu:1347-1378,1606-1634).
 * h_* are HOST arrays: the edge slots (pose index, landmark index, flags; landmark-major) and the
 * Hsc pattern.  Fills the plan fields of *hs (the pattern / contribution-list fields are left
 * alone).  Returns CUGO_ERR_INVALID, with the plan fields cleared, when a landmark's active edges
 * straddle two 256-slot groups: cugo_compute_schur then takes the gather kernels. */
typedef struct cugo_hsc_plan cugo_hsc_plan;
int cugo_hsc_plan_create(cugo_ctx* ctx, int n_edges, int n_poses_free, const int32_t* h_pose,
                         const int32_t* h_lm, const uint8_t* h_flags, const int32_t* h_rowptr,
                         const int32_t* h_colind, cugo_hsc_struct* hs, cugo_hsc_plan** out);
void cugo_hsc_plan_destroy(cugo_hsc_plan* plan);

/* ref: gpu::addLambda(Hll) + gpu::computeBschure + gpu::computeHschure
 * (cuda_block_solver.h:83-109; .cu:1256-1345).  Damping is applied on the fly (Hpp/Hll are
 * left undamped, so no backup/restoreDiagonal pass exists).
 *   d_invHll [Lfree][9] = (Hll + lambda I)^-1,  d_T [E][18] = Hpl * invHll,
 *   d_bsc [Pfree][6] = bp - sum T bl,
 *   d_Hsc [B][36] = Hpp(diag) - sum T Hpl^T            (+ lambda I on diagonal blocks when
 *   damp_hsc_diag != 0, which reproduces the reference's damped Hsc exactly).
 * With a landmark-major plan in *hs (cugo_hsc_plan_create) the products are formed inside the edge
 * pass and d_T may be NULL (T is then never written). */
int cugo_compute_schur(cugo_ctx* ctx, const cugo_edges* ev, const cugo_hsc_struct* hs,
                       double lambda, int damp_hsc_diag, const double* d_Hpp, const double* d_bp,
                       const double* d_Hll, const double* d_bl, const void* d_Hpl,
                       double* d_invHll, void* d_T, double* d_bsc, double* d_Hsc);

/* Sparse block LL^T of Hsc: replaces HscSparseLinearSolver / CuSparseCholeskySolver
 * (ref: src/cuda_linear_solver.cpp:27-57, src/cholesky.hpp:170-309 — cuSOLVER csrchol +
 * METIS).  analyze() = ordering + symbolic (host) once per structure; factor_solve() =
 * numeric multifrontal LL^T + triangular solves on the device per LM trial. */
typedef struct cugo_chol cugo_chol;
int cugo_chol_create(cugo_ctx* ctx, cugo_chol** out);
void cugo_chol_destroy(cugo_chol* s);
/* host pattern: upper block CSR incl. diagonal (ref: initialize(Hsc pattern)) */
int cugo_chol_analyze(cugo_chol* s, int n_block_rows, const int32_t* h_rowptr,
                      const int32_t* h_colind);
/* solve (Hsc + lambda I) x = bsc.  d_Hsc [B][36] upper blocks in the analysed pattern,
 * d_bsc / d_x [n_block_rows][6].  Asynchronous: the zero-pivot flag is written to
 * d_fail[0] (int32 device, 0 = ok) — ref: solve()->bool, zero pivot tol 1e-14. */
int cugo_chol_factor_solve(cugo_chol* s, const double* d_Hsc, double lambda, const double* d_bsc,
                           double* d_x, int32_t* d_fail);
/* statistics of the analysis: nnz(L) in scalars, factorisation flops, #supernodes, #stages */
int cugo_chol_stats(const cugo_chol* s, double* nnzL, double* flops, int* n_supernodes,
                    int* n_stages, double* front_bytes);
/* symbolic plan export (host arrays) so tests can replay the plan with numpy.  A solver
 * created with ctx == NULL is host-only: analyze() builds the plan and skips the upload. */
int cugo_chol_plan_sizes(const cugo_chol* s, int* n, int* n_super, int* n_rows_total);
int cugo_chol_plan_get(const cugo_chol* s, int32_t* perm, int32_t* super_ptr, int32_t* rows_ptr,
                       int32_t* rows, int32_t* parent);
/* named int32 plan array ("perm", "super_ptr", "rows_ptr", "rows", "sparent", "child_ptr",
 * "child", "rel_ptr", "rel", "ncb", "nb", "col0", "col_front", "stage_task_ptr", "task_ptr",
 * "task_fronts", "blk_front", "blk_row", "blk_col", "blk_trans"); pointer valid until the
 * next analyze()/destroy. Returns the length or a negative error. */
int cugo_chol_plan_array(cugo_chol* s, const char* name, const int32_t** out);

/* ref: gpu::schurComplementPost + updatePoses + updateLandmarks + computeScale
 * (cuda_block_solver.h:133-150; .cu:1419-1490).  Reads estimates from d_*_in, writes the
 * updated ("trial") estimates to d_*_out (push/pop of block_solver.cpp:431-439 becomes a
 * buffer swap).  d_scale[0] = sum_i x_i (lambda x_i + b_i) over [xp; xl]. */
int cugo_backsubst_update(cugo_ctx* ctx, const cugo_edges* ev, double lambda,
                          const double* d_invHll, const double* d_bl, const double* d_bp,
                          const void* d_Hpl, const double* d_xp, double* d_xl,
                          const double* d_poses_in, const double* d_lms_in, double* d_poses_out,
                          double* d_lms_out, double* d_scale);

/* ---- (2) graph-level entry points ---------------------------------------------------- */

typedef struct cugo_graph cugo_graph; /* CudaGraphOptimisationImpl + its vertex/edge sets */

/* ref: CudaGraphOptimisationImpl(options) include/cuda_graph_optimisation.h:206;
 * GraphOptimisationOptions src/graph_optimisation_options.h:8-19 */
int cugo_graph_create(int per_edge_information, int per_edge_camera, cugo_graph** out);
/* Extension: a plan-only graph needs no GPU.  cugo_graph_initialize() then runs the host side only —
 * flattening and index assignment (ref: src/block_solver.cpp:21-137), Hsc pattern and product lists
 * (ref: src/sparse_block_matrix.cpp:63-156), ordering + symbolic factorisation (ref:
 * src/cholesky.hpp:97-98,295-296) — and cugo_graph_structure_stats() reports the result;
 * cugo_graph_optimize() fails.  Sizes a problem ahead of time; also what `make SAN=1` exercises. */
int cugo_graph_create_plan_only(int per_edge_information, int per_edge_camera, cugo_graph** out);
void cugo_graph_destroy(cugo_graph* g);
/* vertices: ids are caller ids (ref: PoseVertex(id, Se3D, fixed), LandmarkVertex(id, Vec3d,
 * fixed); src/optimisable_graph.h:109-155) */
int cugo_graph_add_poses(cugo_graph* g, int n, const int32_t* ids, const double* q_t7,
                         const uint8_t* fixed);
int cugo_graph_add_landmarks(cugo_graph* g, int n, const int32_t* ids, const double* xyz,
                             const uint8_t* fixed);
/* edges: dim = 2 -> MonoEdgeSet, 3 -> StereoEdgeSet (ref: include/ba_types.h:34-169,238-254).
 * meas is [n][dim]; info [n]; cam [n][5] or NULL to use the set camera. */
int cugo_graph_add_edges(cugo_graph* g, int dim, int n, const int32_t* pose_ids,
                         const int32_t* landmark_ids, const double* meas, const double* info,
                         const double* cam5);
int cugo_graph_set_camera(cugo_graph* g, int dim, const double* cam5);
int cugo_graph_set_information(cugo_graph* g, int dim, double info);
int cugo_graph_set_robust_kernel(cugo_graph* g, int dim, int type, double delta);
/* multi-GPU: this process handles shard `rank` of `world` (landmark ranges).  exchange() is
 * called on the host with a DEVICE buffer that must be all-reduced in place over all ranks
 * (op 0 = sum, 1 = max) before it returns, or — op >= 2 — overwritten on every rank with rank
 * (op - 2)'s content (a broadcast: the update blocks and solution ranges of the elimination subtrees a
 * rank owns in the sparse LL^T; CUGO_OWN_SUBTREES=0 keeps the factorisation replicated and never asks). */
/* op == -1: a sum REDUCE-SCATTER of `world` equal segments (n_doubles / world doubles each): on return segment `rank`
 * of the buffer must hold the sum over ranks of that segment (the other segments are not read afterwards) — the
 * ownership-keyed exchange of the Schur system: a rank only needs the Hsc blocks its own fronts assemble. */
typedef void (*cugo_exchange_fn)(void* d_buf, size_t n_doubles, int op, void* user);
/* ref: EdgeSet::setOutlierThreshold (src/optimisable_graph.h:737-740) + updateEdges
 * (optimisable_graph.hpp:603-640): at the end of cugo_graph_optimize every edge of the set (dim 2
 * = mono, 3 = stereo) whose chi2 in the last error pass exceeds `threshold` becomes inactive
 * (left out by the next initialize / optimize).  0 disables (default). */
int cugo_graph_set_outlier_threshold(cugo_graph* g, int dim, double threshold);
/* outliers found since the last initialize (ref: getOutlierCount) */
int cugo_graph_n_outliers(cugo_graph* g, int dim);
/* active flag of the first n edges of the set, in insertion order */
int cugo_graph_get_edge_active(cugo_graph* g, int dim, int n, uint8_t* active);
int cugo_graph_set_shard(cugo_graph* g, int rank, int world, cugo_exchange_fn fn, void* user);
/* Native exchange (one process per GPU, RCCL over xGMI; not in the reference, which is single-GPU:
 * src/cuda_device.cpp:264-282).  Rank 0 of the job calls cugo_comm_unique_id() (ncclGetUniqueId,
 * CUGO_UNIQUE_ID_BYTES bytes) and hands the id to every rank by any channel; every rank then calls
 * cugo_comm_create() (a collective: ncclCommInitRank on the current device) once per process and
 * attaches the communicator to its optimisers with cugo_graph_set_comm().  From then on the
 * per-trial sum of [Hsc | bsc] and of (F-hat, scale) are ncclAllReduce calls queued on the
 * solver's own stream: no host synchronisation and no callback.  librccl is loaded on first use.
 * The communicator must outlive the last optimize() of the graphs that use it. */
#define CUGO_UNIQUE_ID_BYTES 128
typedef struct cugo_comm cugo_comm;
int cugo_comm_unique_id(void* id128);
int cugo_comm_create(const void* id128, int rank, int world, cugo_comm** out);
void cugo_comm_destroy(cugo_comm* comm);
int cugo_graph_set_comm(cugo_graph* g, cugo_comm* comm);
/* payload bytes and number of all-reduce calls since the last cugo_graph_initialize() */
int cugo_graph_exchange_stats(cugo_graph* g, double* bytes, int32_t* calls);
/* hipSetDevice for the calling thread: a rank binds its GPU (LOCAL_RANK) before creating graphs */
int cugo_set_device(int device);
/* the landmark index range [*l0, *l1) that shard `rank` of `world` owns, given the number
 * of active edges of every landmark (host only; the rule cugo_graph_initialize applies) */
int cugo_shard_range(int n_landmarks_total, const int32_t* edges_per_landmark, int rank, int world,
                     int* l0, int* l1);
/* ref: initialize() :147.  When nothing but vertex estimates changed since the last call (no vertex /
 * edge added or removed, no fixed flag, measurement, information, camera, robust kernel or threshold
 * touched) the flattened graph on the device is kept and only the estimates are uploaded; the
 * counter below says how often that happened (CUGO_NO_FLATTEN_REUSE=1 disables it). */
int cugo_graph_initialize(cugo_graph* g);
int cugo_graph_flatten_reuses(cugo_graph* g);
/* Run-time switches of one optimiser.  Every optimiser takes a snapshot of the CUGO_* environment variables when it
 * is created (README.md lists them; nothing is read per call); this changes a single switch afterwards:
 *   "flatten_reuse"   (0 = CUGO_NO_FLATTEN_REUSE: flatten and upload the whole graph in every initialize)
 *   "structure_reuse" (0 = CUGO_NO_STRUCTURE_REUSE: Hsc pattern, lists, ordering, symbolic factor rebuilt every time —
 *                      what a caller that builds a new graph per call, e.g. ORB-SLAM2, pays)
 *   "init_timing"     (1 = CUGO_INIT_TIMING: per-section host times of initialize / optimize on stderr)
 * Returns CUGO_ERR_INVALID for an unknown name. */
int cugo_graph_set_option(cugo_graph* g, const char* name, int value);
int cugo_graph_optimize(cugo_graph* g, int n_iters);  /* ref: optimize(n)  :153 */
int cugo_graph_n_stats(cugo_graph* g);                /* ref: batchStatistics() :158 */
int cugo_graph_get_stats(cugo_graph* g, int32_t* iteration, double* chi2, int cap);
int cugo_graph_get_trace(cugo_graph* g, double* lambda, double* rho, int32_t* trials, int cap);
int cugo_graph_get_poses(cugo_graph* g, int n, const int32_t* ids, double* q_t7);
int cugo_graph_get_landmarks(cugo_graph* g, int n, const int32_t* ids, double* xyz);
/* overwrite estimates of existing vertices (ref: Vertex::setEstimate,
 * src/optimisable_graph.h:128); takes effect at the next cugo_graph_initialize() */
int cugo_graph_set_poses(cugo_graph* g, int n, const int32_t* ids, const double* q_t7);
int cugo_graph_set_landmarks(cugo_graph* g, int n, const int32_t* ids, const double* xyz);
int cugo_graph_n_active_edges(cugo_graph* g);
/* per-phase milliseconds accumulated since initialize(): ref getTimeProfile
 * (block_solver.cpp:470-488).  names is a '\n' separated list written into buf. */
int cugo_graph_time_profile(cugo_graph* g, char* names_buf, int buf_len, double* ms, int cap);
int cugo_graph_set_verbose(cugo_graph* g, int verbose);
/* HIP-event timing on the solver's own stream (diagnostic).  on = 1: an event pair round every kernel group (build,
 * errors, schur, cholesky, backsubst_update, exchange) and every kernel — per-kernel figures; a pair brackets the kernel
 * and its dispatch, 1 - 2 us more than the kernel's own duration, so these are never to be summed; on = 2: ONE event per group boundary, so that the group times add up exactly to the device time
 * between the first and the last event of cugo_graph_optimize; 0: off.  names: '\n' separated. */
int cugo_graph_set_kernel_timing(cugo_graph* g, int on);
/* fp32-internal mode (ref: the USE_FLOAT32 build option, CMakeLists.txt:8; here a run-time switch,
 * GraphOptimisationOptions::useFloat32): float storage of the Hpl / Hpl*Hll^-1 block streams,
 * everything else fp64.  Takes effect at the next cugo_graph_initialize(). */
int cugo_graph_set_float32(cugo_graph* g, int on);
int cugo_graph_kernel_times(cugo_graph* g, char* names_buf, int buf_len, double* ms,
                            int32_t* launches, int cap);
/* device-to-device copy on the context stream (used by exchange callbacks) */
int cugo_memcpy_d2d(cugo_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
/* solver statistics of the last buildStructure, in this order: B (Hsc blocks), M (block
 * products), nnz(L), Cholesky flops, supernodes, stages, front bytes, off-diagonal products,
 * then the algorithmic work of the batched Cholesky kernels per factorisation: potrf flops,
 * trsm flops, syrk flops, extend-add bytes, backward bytes.  Returns the count written. */
int cugo_graph_structure_stats(cugo_graph* g, double* out, int cap);

/* seeded ORB-SLAM-style synthetic graph built directly into a graph.
 * Returns arrays through the getters above; also usable to feed the CPU oracle. */
typedef struct cugo_synth_params
{
    int n_poses, n_landmarks, n_edges; /* exact counts */
    double stereo_fraction;            /* fraction of landmarks observed in stereo */
    double pixel_noise, pose_rot_noise, pose_trans_noise, landmark_noise_rel;
    int n_loop_closures;
    uint64_t seed;
} cugo_synth_params;
/* fills caller-provided arrays (sizes from params): poses [P][7], lms [L][3],
 * e_pose/e_lm [E], e_stereo [E], e_meas [E][3], e_omega [E]; cam5 [5].  ids == positions;
 * pose 0 is the gauge (fixed). */
int cugo_synth_generate(const cugo_synth_params* p, double* poses, double* lms, int32_t* e_pose,
                        int32_t* e_lm, uint8_t* e_stereo, double* e_meas, double* e_omega,
                        double* cam5);

#ifdef __cplusplus
}
#endif
#endif /* CUGO_HIP_H */
