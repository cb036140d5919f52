// Host-side ordering + symbolic analysis for the multifrontal block LL^T.
// Replaces cusolverSpXcsrmetisndHost + csrcholAnalysis (ref: src/cholesky.hpp:97-98,295-296).
#pragma once
#include <cstdint>
#include <vector>

namespace cugo_host
{

struct CholOptions
{
    int nd_leaf = 96;        // nested dissection stops below this many block nodes (swept on MI355X: tools/sweep_ordering.sh)
    int max_super_cols = 16; // relaxed supernodes: at most this many block columns
    double zero_frac = 0.35; // relaxed supernodes: tolerated share of explicit zero blocks
    int target_tasks = 1024; // subtree-to-workgroup granularity of stage 0
    bool alias_chains = true; // single-child fronts with identical rows live in the child's update block
    // bottom subtrees walked by one workgroup each (k_subtree_factor) only if there are at least
    // this many of them.  Measured on MI355X the batched per-level kernels win at every size tried
    // (1.3k poses: 13 subtrees; 10k poses: ~900 fronts, 56.4 vs 57.7 ms), so it is off by default;
    // CUGO_MIN_SUBTREE_TASKS=0 turns it on (kept under test).
    int min_subtree_tasks = 1 << 30;
    int max_front_cols = 16; // hard cap on pivot block columns of a front (LDS-resident L11)
    // a level with more 64x64 tiles than this solves its L21 row tiles once (k_up_trsm) and then runs
    // syrk-only tiles (k_up_syrk) instead of the fused trsm+syrk tile kernel (0: never)
    int two_phase_min_tiles = 260; // (re-swept in round 3 with the 16-column potrf: 128 / 260 / 400 / 600 / never = 11.13 / 11.00 / 11.08 /
                                   // 11.29 / 11.27 ms per step on the kitti_00 shape; 128 / 400 / 800 / 1600 / never = 38.8 / 38.9 / 38.0 / 39.4 / 40.7 ms on the 10k graph)
    bool xcd_affinity = true; // tile items of a front share an index class mod 8, i.e. an XCD and its L2 (CUGO_XCD_AFFINITY=0: listed front by front)
    int tile32_max_tiles = 64; // a level with at most this many 64x64 tiles is cut into 32x32 tiles (0: never)
    // landmark-sharded run (one process per GPU): this rank's schedule holds the fronts of the elimination
    // subtrees it OWNS plus the replicated top of the tree (CholPlan::owner); world == 1: everything
    int rank = 0, world = 1;
    // the rank-owned form (ownership, update-block and solution-range broadcasts) — normally world > 1; it can be
    // forced for world == 1 (everything below the top belongs to rank 0): the one-GPU rehearsal of the form
    bool owned = false;
    static CholOptions from_env();
};

struct CholPlan
{
    int n = 0; // block rows/cols of the matrix
    std::vector<int32_t> perm, iperm; // perm[new] = old

    int n_super = 0;
    std::vector<int32_t> super_ptr;      // [n_super+1] column ranges (new ordering)
    std::vector<int32_t> rows_ptr, rows; // boundary block rows per supernode (new ordering)
    std::vector<int32_t> sparent;        // parent supernode, -1 for roots
    std::vector<int32_t> child_ptr, child;
    std::vector<int32_t> rel_ptr, rel;   // per (child) supernode: block row in the parent front
    std::vector<int32_t> ncb, nb, col0;
    std::vector<int64_t> off;
    int64_t front_doubles = 0;
    // leading dimension of each front and, for a front stored inside the update block of its only
    // child (identical row structure: the extend-add would be a plain copy), that child; else -1
    std::vector<int64_t> ldf;
    std::vector<int32_t> alias_of;
    int n_aliased = 0;
    std::vector<int64_t> woff; // per front: offset of W = L11^-1 (pad16(6*ncb)^2 doubles)
    int64_t winv_doubles = 0;
    // per front of an upper stage: offset of its L21 (+ forward-solved rhs row) in the compact
    // l21 buffer, column-major with leading dimension 6*(nb-ncb)+1; -1 for subtree-stage fronts
    // (their L21 stays in the front matrix)
    std::vector<int64_t> l21off;
    int64_t l21_doubles = 0;
    std::vector<int32_t> col_front;      // new column -> supernode

    int n_stages = 0;
    std::vector<int32_t> stage_task_ptr; // [n_stages+1]
    std::vector<int32_t> task_ptr, task_fronts;
    bool has_subtree_stage = false; // stage 0 = multi-front subtree tasks (one workgroup each)
    // upper stages: every task is one front processed by four batched kernels; their work
    // items are triples (front, a, b) in `wl`:
    //   extend-add  : parent block columns [a, b)
    //   trsm        : scalar rows [a, b) below the pivot block (relative to row 6*ncb)
    //   syrk        : 64x64 tile (row tile a, col tile b), a >= b
    std::vector<int32_t> wl;
    std::vector<int32_t> ea_ptr, eab_ptr, syrk_ptr, bwg_ptr; // [n_stages+1] item ranges per stage
    // look-ahead schedule (chol_kernels.hip, k_up_potrf_la / k_up_lead): la_np = lead rows of a front
    // (leading boundary block rows inside the parent's pivot columns), items lead: (front, -, -) per
    // front with lead rows, sb: the syrk tiles that are not wholly inside the lead block
    int clr0 = 0, nclr = 0; // items (front, first column, past-last column) of the lower-triangle clear
    // assembly that leaves nothing to clear (k_assemble_fronts): items (front, first block column, past-last
    // block column) over the fronts with storage of their own; asm_map at asm_off[front] (-1: no storage):
    // [nb entries: the permuted block column whose right-hand side sits in this column's rhs row, or -1]
    // [packed lower triangle of blocks, column-major: 2 * Hsc block + transposed, or -1 = fill]
    int asm0 = 0, nasm = 0;
    std::vector<int32_t> asm_map;
    std::vector<int64_t> asm_off;
    std::vector<int32_t> la_np;
    std::vector<int32_t> lead_ptr, sb_ptr;
    // edge of the syrk tiles of each stage: 64, or 32 where a level has so few 64-tiles that the
    // launch would leave most CUs idle (its duration is then one tile's, and a 32-tile is shorter)
    std::vector<int32_t> stage_tile; // (0: the two-phase form, items trsm_ptr / syrk_ptr)
    std::vector<int32_t> trsm_ptr;   // [n_stages+1] trsm items (front, first row below the pivots, rows)
    // backward pass: boundary block rows of a front that lie in its PARENT's pivot columns (they come
    // first).  The mat-vec over the remaining rows — ancestors above the parent, solved earlier —
    // is done one launch ahead by extra workgroups riding with the parent's level (items bwg:
    // front, first column, -); -1: the front does its whole mat-vec itself
    std::vector<int32_t> bw_np;
    // ea = extend-add of the pivot block columns (before potrf), eab = of the boundary columns
    // (same launch as trsm)
    int nc_max = 6; // widest pivot block in scalars (LDS sizing)

    // ---- rank-owned elimination subtrees (world > 1) -------------------------------------------------
    // owner[f] = rank that factors front f, or -1: the front belongs to the replicated top of the tree
    // (every rank factors it).  Proportional mapping: the ranks of a node are split over its children by
    // subtree work; a node whose range is down to one rank is owned, subtree and all, by that rank.
    // The schedule of THIS rank (task lists, work items) holds only the fronts it owns or replicates.
    std::vector<int32_t> owner;
    // update blocks that cross the ownership boundary: fronts (front, stage, owner) owned by one rank whose
    // parent is replicated — their update block (columns 6 ncb .. of the front, contiguous) is broadcast from
    // the owner after `stage` so that every rank can run the parent's extend-add
    std::vector<int32_t> xu_front, xu_stage, xu_owner;
    // solution ranges that cross it: block columns [lo, hi) (new ordering) solved by `owner` alone
    std::vector<int32_t> xx_lo, xx_hi, xx_owner;
    double rank_flops = 0, top_flops = 0; // factorisation work of this rank's own subtrees / of the replicated top
    bool owned = false; // this plan is in the rank-owned form (CholOptions::owned)
    // Ownership-keyed exchange of the Schur system (owned form): a rank only needs the SUM over ranks of the Hsc
    // blocks and right-hand-side rows that are assembled into fronts it factors — its own subtrees and the
    // replicated top.  Packed buffer: `world` segments of xs_seg doubles (segment r: the units of rank r's fronts,
    // blocks first, padded to the longest segment) followed by xs_top doubles (the top's units); the segments are
    // reduce-scattered (rank r receives segment r), the top part is all-reduced.  xs_off[u]: offset of unit u in
    // that buffer — u < B: Hsc block u (36 doubles), u >= B: the right-hand-side row of pose u - B (6 doubles).
    int64_t xs_seg = 0, xs_top = 0;
    std::vector<int64_t> xs_off;

    std::vector<int32_t> blk_front, blk_row, blk_col; // per Hsc block
    std::vector<uint8_t> blk_trans;

    long ld_max = 1;
    double nnzL = 0;  // scalars in L (incl. explicit zeros of relaxed supernodes)
    double flops = 0; // factorisation flops (2 * multiply-adds)
    // algorithmic work of the batched upper-stage kernels (per factorisation)
    double up_potrf_flops = 0, up_trsm_flops = 0, up_syrk_flops = 0, up_ea_bytes = 0, backward_bytes = 0;
};

// pattern: upper block CSR (columns ascending, diagonal included)
void chol_analyze(int n, const int32_t* rowptr, const int32_t* colind, const CholOptions& opt,
                  CholPlan& plan);

} // namespace cugo_host
