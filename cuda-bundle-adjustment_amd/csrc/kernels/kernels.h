// Host-callable launchers of the HIP kernels (internal; the public surface is include/cugo_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../../include/cugo_hip.h"

namespace cugo_k
{

// Optional per-kernel timing hook: when installed, every kernel launch of the library is
// bracketed by begin(name)/end(name) (the engine records HIP events on the launch stream).
struct LaunchHook
{
    virtual void begin(const char* kernel, hipStream_t s) = 0;
    virtual void end(const char* kernel, hipStream_t s) = 0;
    virtual ~LaunchHook() {}
};
void set_launch_hook(LaunchHook* h); // nullptr = off (default)
LaunchHook* launch_hook();
struct LaunchScope
{
    const char* name;
    hipStream_t s;
    LaunchHook* h;
    LaunchScope(const char* n, hipStream_t st) : name(n), s(st), h(launch_hook())
    {
        if (h)
            h->begin(name, s);
    }
    ~LaunchScope()
    {
        if (h)
            h->end(name, s);
    }
};
// launch + per-kernel timing scope
#define CUGO_LAUNCH(kernel, grid, block, lds, stream, ...)                          \
    do                                                                              \
    {                                                                               \
        ::cugo_k::LaunchScope _scope(#kernel, stream);                              \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);          \
    } while (0)

// scratch for deterministic two-stage reductions: partial sums per workgroup
struct ReduceScratch
{
    double* d_partials; // capacity doubles
    size_t capacity;
};

// --- edge / landmark / pose passes (ba_kernels.hip) ---------------------------------------
void launch_errors(hipStream_t s, const cugo_edges& ev, const double* d_poses, const double* d_lms,
                   cugo_robust rk, ReduceScratch rs, double* d_chi);

// d_Hpl / d_T below: double [E][18], or float [E][18] when ev.block_f32 is set
// fuse_lambda >= 0 with d_invHll and d_T given: the pass also leaves invHll = (Hll + lambda I)^-1 and
// T = Hpl invHll (what launch_schur's edge kernel computes; pass have_T = true there).  Only for slot
// layouts that keep every landmark's edges inside one 256-slot group (the engine's).
// With d_lmrec as well (16 doubles per free landmark) the pass writes ONE block stream instead: G = Hpl L^-T into d_T
// (Hll + lambda I = L L^T), the line {L^-1 (6), L^-1 bl (3)} of every landmark into d_lmrec, and neither d_Hpl nor
// d_invHll; skip_poses then also leaves out the pose pass (d_Hpp, d_bp).  launch_schur and launch_backsubst_update take
// the same d_lmrec and d_T for that form (SchurRows::d_lmrec).
// chi_behind_scale: the chi2 partials of the pass go behind the scale partials of the update pass in the scratch, for
// launch_trial_tail_from_build.
void launch_build(hipStream_t s, const cugo_edges& ev, const double* d_poses, const double* d_lms,
                  cugo_robust rk, double* d_Hpp, double* d_bp, double* d_Hll, double* d_bl,
                  void* d_Hpl, ReduceScratch rs, double* d_chi, double fuse_lambda = -1.0,
                  double* d_invHll = nullptr, void* d_T = nullptr, double* d_lmrec = nullptr, bool skip_poses = false,
                  bool chi_behind_scale = false);

// the reductions that end a trial, with the chi2 partials of a build pass queued with chi_behind_scale (ba_kernels.hip)
void launch_trial_tail_from_build(hipStream_t s, const cugo_edges& ev, ReduceScratch rs, int n_scale_partials,
                                  double* d_out, const double* d_flag, double* h_out, double seq, unsigned* d_done);

void launch_max_diagonal(hipStream_t s, const double* d_Hpp, int nP, const double* d_Hll, int nL,
                         ReduceScratch rs, double* d_out);

// SchurRows (optional): the per-entry records of the pose-major edge list (launch_pose_rec) and the longest
// block row of Hsc.  Given, the H-side and b-side of the Schur complement run as ONE kernel that forms the
// whole block row of a pose (k_hsc_rows): T = Hpl invHll is never written or read (d_T may be NULL) and the
// product lists of hs are not used; have_T then means "d_invHll already holds (Hll + lambda I)^-1".
struct SchurRows
{
    const int32_t* d_pose_rec = nullptr; // [n][4]: slot, end of its landmark's slots, landmark, flags
    int max_row_nnz = 0;
    // row-strip form of the gather kernels (k_hsc_offdiag_strip): per product, the position of its T edge in
    // the edge list of its pose (launch_list_pos).  Given, the off-diagonal blocks are formed one block row per
    // workgroup with the row's T blocks staged in LDS once; same sums, bit for bit
    const int32_t* d_off_pi = nullptr;
    // which form of the H-side gather kernels (Options::hsc_mfma / hsc_xcd, read once per context / optimiser):
    // 1 both on the matrix cores, 2 only the off-diagonal one, 0 the vector-lane kernels; blocks dealt to the XCDs
    // in contiguous ranges or in dispatch order
    int mfma = 1;
    bool xcd = true;
    // fused iteration in the one-stream form (launch_build with d_lmrec and skip_poses for this lambda: d_T holds G):
    // the off-diagonal kernel takes both operands from d_T, and k_pose_schur forms the diagonal blocks, bp (d_bp_out) and
    // bsc from the build pass's records, the landmarks' lines and the poses the pass linearised at (d_poses)
    const double* d_lmrec = nullptr;
    const double* d_poses = nullptr;
    ReduceScratch rs{nullptr, 0};
    double* d_bp_out = nullptr;
};
// d_pose_pos [n_edges] scratch/out: position of every slot in its pose's list; d_off_pi [M] out
void launch_list_pos(hipStream_t s, const cugo_edges& ev, int n_list, size_t M, const int32_t* d_off_ei,
                     int32_t* d_pose_pos, int32_t* d_off_pi);
void launch_pose_rec(hipStream_t s, const cugo_edges& ev, int n, int32_t* d_rec);
// false: the row kernel cannot take this structure (no pattern on the device, or a block row too long for
// its LDS accumulators): launch_schur then runs the gather kernels, which need d_T
bool schur_rows_usable(const cugo_hsc_struct& hs, int max_row_nnz);
void launch_schur(hipStream_t s, const cugo_edges& ev, const cugo_hsc_struct& hs, double lambda,
                  int damp_hsc_diag, const double* d_Hpp, const double* d_bp, const double* d_Hll,
                  const double* d_bl, const void* d_Hpl, double* d_invHll, void* d_T,
                  double* d_bsc, double* d_Hsc, bool have_T = false, SchurRows rows = SchurRows());

// lambda_pose: damping used in the pose part of the scale sum (0 on ranks > 0 of a sharded run
// so that the all-reduced scale counts lambda*|xp|^2 once)
// d_scale == nullptr: the scale partials are left in the scratch (launch_errors_tail sums them);
// returns their number
int launch_backsubst_update(hipStream_t s, const cugo_edges& ev, double lambda, double lambda_pose,
                            const double* d_invHll, const double* d_bl, const double* d_bp,
                            const void* d_Hpl, const double* d_xp, double* d_xl,
                            const double* d_poses_in, const double* d_lms_in, double* d_poses_out,
                            double* d_lms_out, ReduceScratch rs, double* d_scale,
                            const double* d_lmrec = nullptr); // d_lmrec: d_Hpl holds G, see k_backsubst_landmarks
// error pass of an LM trial + BOTH final reductions in one launch: d_out[0] = chi2, d_out[1] = scale
// (from the n_scale_partials left by launch_backsubst_update); h_out (pinned host memory, may be
// null) receives {chi2, scale, the 8 bytes at d_flag}: readable after the stream has been waited for
void launch_errors_tail(hipStream_t s, const cugo_edges& ev, const double* d_poses, const double* d_lms,
                        cugo_robust rk, ReduceScratch rs, int n_scale_partials, double* d_out,
                        const double* d_flag, double* h_out, double seq, unsigned* d_done);

size_t reduce_scratch_doubles(int n_edges, int n_poses, int n_landmarks);

// chi_e per edge slot (outlier rejection, ref: computeOutliersKernel cuda_block_solver.cu:1135)
void launch_edge_chi(hipStream_t s, const cugo_edges& e, const double* d_poses, const double* d_lms,
                     cugo_robust rk, double* d_chi_e);

// --- multifrontal LL^T (chol_kernels.hip) -------------------------------------------------
// Device-side plan; all index arrays in units of 6x6 blocks unless noted.
constexpr int TMETA = 36; // ints per task record (CholPlanDev::tmeta)
struct CholPlanDev
{
    int n_fronts;
    // per front
    const int32_t* ncb;        // pivot block columns
    const int32_t* nb;         // total block rows (pivot + boundary); the rhs row is row 6*nb
    const int64_t* off;        // offset (doubles) of the front matrix in `fronts`
    const int64_t* ldf;        // leading dimension: 6*nb + 1, or the child's when the front lives in
                               // the update block of its only child (single-child chains)
    const int32_t* alias_of;   // that child, or -1
    const int32_t* bw_np;      // backward: leading boundary block rows owned by the parent when the
                               // rest of the front's mat-vec is done ahead of time, else -1
    const int32_t* la_np;      // look-ahead: leading boundary block rows inside the parent's pivot columns
                               // (the front's "lead rows"; 0 for a root)
    const int64_t* woff;       // offset (doubles) of W = L11^-1 (pad16(6*ncb)^2, column-major) in winv
    double* winv;
    const int64_t* l21off;     // offset of the front's L21 (+ rhs row) in l21, or -1: in the front itself
    double* l21;               // column-major, leading dimension 6*(nb-ncb)+1
    int nc_max;                // widest pivot block (scalars)
    int ea_lds;                // potrf: children's contributions to F11 go straight into its LDS copy
    int panel16;               // potrf: 16-column register panels (CUGO_PANEL16=0: the 6-column LDS panels)
#ifdef CUGO_DEBUG_HOOKS // (make HOOKS=1 -> libcugo_hip_hooks.so; the product build carries none of this)
    int kernel_acquire;        // CUGO_KERNEL_ACQUIRE: bit 0 = every kernel of the factorisation starts with an agent-scope acquire fence, bit 1 = ends with a release fence
    int dbg_delay;             // diagnosis (CUGO_DEBUG_DELAY): which waves / workgroups of the factorisation's kernels sleep (chol_kernels.hip: dbg_sleep)
    int zero_lds;              // diagnosis (CUGO_DEBUG_ZERO_LDS=1 / 2): every kernel fills its LDS with zeros / NaNs first
    int lds_doubles;           // (set per launch: the dynamic LDS of this launch, for that fill)
    int dbg_skip_wg;           // (set per launch, -1: none) fault injection: this workgroup returns at once (CUGO_DEBUG_SKIP)
#endif
    const int32_t* col0;       // first pivot column (new ordering, block units)
    const int32_t* rows_ptr;   // [n_fronts+1] into rows: boundary block rows (new ordering)
    const int32_t* rows;
    const int32_t* child_ptr;  // [n_fronts+1] children of each front
    const int32_t* child;      // child front ids
    const int32_t* rel_ptr;    // [n_fronts+1] (indexed by CHILD front) into rel
    const int32_t* rel;        // position (block row in the parent front) of each boundary row
    // schedule
    int n_stages;
    const int32_t* ea1;        // per child link: {child, its boundary block rows, its leading rows inside the
                               // parent's pivots, offset of its rel list, update-block offset (int64), its leading
                               // dimension (int64)} — the potrf workgroup's extend-add into F11 (tmeta[16..17])
    const int32_t* tmeta;      // [n_tasks_total][TMETA] (16..17: range in ea1, 18..19 unused): {fronts in the task, first front, its ncb, nb, col0, bw_np,
                               // rows_ptr, has-children-to-add flag, off, ldf, woff, l21off (four int64)} (potrf, backward substitution);
                               // 20..35: the front's first 16 boundary block rows (those inside its parent's pivot block: the
                               // backward substitution gathers x_R through them without a round trip to the row lists)
    // k_assemble_fronts (chol_symbolic.h: CholPlan::asm_map): per stored front its map at asm_off[front]
    const int32_t* asm_map;
    const int64_t* asm_off;
    const int32_t* wl_base;    // the work-item triples (chol_symbolic.h: CholPlan::wl) ...
    const int32_t* fat;        // ... and one 64-byte record per item for the tile kernels (TileItem)
    const int32_t* task_ptr;   // [n_tasks_total+1] into task_fronts
    const int32_t* task_fronts;
    // assembly of A (one entry per Hsc block)
    int n_hsc_blocks;
    const int32_t* blk_front;
    const int32_t* blk_row;    // block row in the front
    const int32_t* blk_col;    // block col in the front
    const uint8_t* blk_trans;  // 1: store transposed
    // rhs / solution mapping
    int n;                     // block rows of the matrix
    const int32_t* perm;       // new -> old
    const int32_t* col_front;  // new col -> front
    // sink for masked-off lanes: lets conditional stores/loads be emitted branch-free (a branch
    // per store makes the compiler wait vmcnt(0) before each one, serialising the stores)
    double* junk;              // 64 x 1024 doubles
};

// clears the fronts (their lower triangles through the nclear items front / first / past-last column;
// nclear == 0: the whole buffer), scatters Hsc (+lambda) and bsc into them and resets *d_fail
void launch_chol_assemble(hipStream_t s, const CholPlanDev& p, double* d_fronts, size_t front_doubles,
                          const double* d_Hsc, double lambda, const double* d_bsc, int32_t* d_fail,
                          const int32_t* d_clear_items, int nclear, const int32_t* d_asm_items = nullptr,
                          int nasm = 0);
void launch_chol_subtree_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                               int ntasks, size_t lds_bytes, int32_t* d_fail);
// one etree level: extend-add(pivot columns) / potrf (+ extend-add of the boundary columns) /
// fused trsm+syrk kernels over the work items d_wl[...]
// tile: edge of the update-matrix tiles of this level's items (64, or 32 on levels with few fronts)
void launch_chol_upper_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                             int ntasks, const int32_t* d_wl, int eap0, int neap, int ea0, int nea,
                             int sy0, int nsy, int tile, size_t lds_bytes, int32_t* d_fail,
                             double* dbg_line = nullptr, double* dbg_scratch = nullptr);
#ifdef CUGO_DEBUG_HOOKS
// diagnosis (CUGO_DEBUG_STALE): exchanges the 16 doubles at `line` with those at `scratch`
void launch_swap16(hipStream_t s, double* line, double* scratch);
#endif
// two-phase form of a level's tile work (stage_tile == 0): trsm items (front, first row below the pivots,
// rows) then syrk items (front, linear tile index, tile columns)
void launch_chol_two_phase(hipStream_t s, const CholPlanDev& p, double* d_fronts, const int32_t* d_trsm, int ntrsm,
                           const int32_t* d_syrk, int nsyrk);
// look-ahead schedule, two launches per level (see k_up_potrf_la / k_up_lead):
//   potrf of level k (ntasks fronts from task0) together with the non-lead update tiles of level k-1
//   (items front, ti, tj of edge `tile`); then the lead workgroups of level k (items front,-,-)
//   together with the extend-add below the lead rows (items front, first, past-last block column)
void launch_chol_potrf_la(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0, int ntasks,
                          const int32_t* d_tiles, int ntiles, int tile, int32_t* d_fail);
void launch_chol_lead(hipStream_t s, const CholPlanDev& p, double* d_fronts, const int32_t* d_lead,
                      int nlead, const int32_t* d_eap, int neap, const int32_t* d_eab, int neab);
// backward substitution of one level: ntasks workgroups solve the level's fronts; ngemv more
// workgroups (items d_wl_gemv: front, first column, -) do the ancestor part of the mat-vec of
// the CHILDREN of these fronts, which the next launch then does not have to wait for
void launch_chol_backward_stage(hipStream_t s, const CholPlanDev& p, double* d_fronts, int task0,
                                int ntasks, size_t lds_bytes, double* d_xnew, double* d_x,
                                const int32_t* d_wl_gemv, int ngemv);
#ifdef CUGO_DEBUG_HOOKS
void launch_nop(hipStream_t s); // diagnosis (CUGO_DEBUG_GAP): an empty kernel
// fault injection (CUGO_DEBUG_SKIP): the launches of the factorisation that follows are counted from 0; workgroup
// target_wg of launch target_launch returns at once; dump_path: the launch table (index, kernel, grid, first items) goes there
void chol_dbg_skip_begin(int target_launch, int target_wg, const char* dump_path);
void chol_dbg_skip_end();
// diagnosis (CUGO_DEBUG_HASH): *out += the sum of the n 64-bit words at p (integer sum: order-independent)
void launch_hash_words(hipStream_t s, const void* p, size_t n_words, unsigned long long* out);
#endif
void launch_flag_to_double(hipStream_t s, int32_t* d_flag); // int32 0 / 1 -> double 0.0 / 1.0 in the same 8-byte slot
void launch_chol_unpermute(hipStream_t s, const CholPlanDev& p, const double* d_xnew, double* d_x);
// ownership-keyed exchange of [Hsc | bsc] (chol_symbolic.h: CholPlan::xs_off; d_off [B + n] packed offsets):
// every unit of d_sys into the packed buffer / the units at packed offsets [lo, hi) or >= top0 back into d_sys
void launch_xs_pack(hipStream_t s, const int64_t* d_off, int B, int n, const double* d_sys, double* d_xbuf);
void launch_xs_unpack(hipStream_t s, const int64_t* d_off, int B, int n, const double* d_xbuf, double* d_sys, long lo,
                      long hi, long top0);
void set_debug_stamps(long long* d_buf); // diagnostic s_memtime stamps (nullptr = off)
size_t chol_lds_factor_bytes(int nc_max);
size_t chol_lds_backward_bytes(int nc_max, long ld_max);
int chol_max_pivot_cols(); // widest pivot block the kernels support (scalars)

} // namespace cugo_k
