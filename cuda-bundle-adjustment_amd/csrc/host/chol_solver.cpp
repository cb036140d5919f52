// cugo_chol_*: sparse block LL^T solver object of the C ABI (include/cugo_hip.h).
// ref: HscSparseLinearSolver (src/cuda_linear_solver.cpp:27-57) + CuSparseCholeskySolver
// (src/cholesky.hpp:170-309): initialize(pattern) once, solve(A, b, x)->bool per LM trial.
#include "chol_solver.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

using namespace cugo_host;

// host half of upload(): the plan's index arrays and per-task / per-item records packed into two vectors (done
// by analyze_host, i.e. beside initialize() when the structure is analysed on the helper thread)
void cugo_chol::pack()
{
    const CholPlan& P = plan;
    // All index arrays of the plan travel in TWO host-to-device copies (a pageable copy costs
    // ~20 us whatever its size, and there are 23 arrays: 0.4 ms of a cold call on a small graph)
    pack32.clear(), pack64.clear();
    auto put32 = [this](const std::vector<int32_t>& v) {
        const size_t o = pack32.size();
        pack32.insert(pack32.end(), v.begin(), v.end());
        pack32.resize((pack32.size() + 3) & ~size_t(3)); // keep every array 16-byte aligned
        return o;
    };
    auto put64 = [this](const std::vector<int64_t>& v) {
        const size_t o = pack64.size();
        pack64.insert(pack64.end(), v.begin(), v.end());
        pack64.resize((pack64.size() + 1) & ~size_t(1));
        return o;
    };
    const size_t o_ncb = put32(P.ncb), o_nb = put32(P.nb), o_col0 = put32(P.col0);
    const size_t o_alias = put32(P.alias_of), o_bwnp = put32(P.bw_np), o_lanp = put32(P.la_np);
    const size_t o_rows_ptr = put32(P.rows_ptr), o_rows = put32(P.rows);
    const size_t o_child_ptr = put32(P.child_ptr), o_child = put32(P.child);
    const size_t o_rel_ptr = put32(P.rel_ptr), o_rel = put32(P.rel);
    const size_t o_task_ptr = put32(P.task_ptr), o_task_fronts = put32(P.task_fronts);
    const size_t o_blk_front = put32(P.blk_front), o_blk_row = put32(P.blk_row), o_blk_col = put32(P.blk_col);
    const size_t o_perm = put32(P.perm), o_col_front = put32(P.col_front), o_wl = put32(P.wl);
    // one 64-byte record per task (kernels.h: CholPlanDev::tmeta)
    const size_t ntask = P.task_ptr.empty() ? 0 : P.task_ptr.size() - 1; // (no free pose: an empty plan)
    // child records of the extend-add into F11 (kernels.h: CholPlanDev::ea1), per front in child order
    std::vector<int32_t> ea1, ea1_ptr(P.n_super + 1, 0);
    for (int f = 0; f < P.n_super; f++)
    {
        if (P.alias_of[f] < 0)
            for (int k = P.child_ptr[f]; k < P.child_ptr[f + 1]; k++)
            {
                const int c = P.child[k];
                const int64_t ncs_c = 6 * (int64_t)P.ncb[c], ldc = P.ldf[c];
                const int64_t q[2] = {P.off[c] + ncs_c * ldc + ncs_c, ldc};
                int32_t r[8] = {c, P.nb[c] - P.ncb[c], P.la_np[c], P.rel_ptr[c], 0, 0, 0, 0};
                std::memcpy(r + 4, q, sizeof q);
                ea1.insert(ea1.end(), r, r + 8);
            }
        ea1_ptr[f + 1] = (int32_t)(ea1.size() / 8);
    }
    ea1.resize(ea1.size() + 8, 0);
    const size_t o_ea1 = put32(ea1);
    std::vector<int32_t> tmeta(cugo_k::TMETA * ntask, 0);
    for (size_t t = 0; t < ntask; t++)
    {
        int32_t* m = tmeta.data() + cugo_k::TMETA * t;
        const int f = P.task_fronts[P.task_ptr[t]];
        m[0] = P.task_ptr[t + 1] - P.task_ptr[t], m[1] = f, m[2] = P.ncb[f], m[3] = P.nb[f], m[4] = P.col0[f];
        m[5] = P.bw_np[f], m[6] = P.rows_ptr[f];
        m[7] = (P.alias_of[f] < 0 && P.child_ptr[f + 1] > P.child_ptr[f]) ? 1 : 0;
        m[16] = ea1_ptr[f], m[17] = ea1_ptr[f + 1];
        for (int q = 0; q < 16; q++)
        { // the first boundary block rows (clamped: a front with fewer repeats its last one)
            const int nrows = P.rows_ptr[f + 1] - P.rows_ptr[f];
            m[20 + q] = nrows > 0 ? P.rows[P.rows_ptr[f] + std::min(q, nrows - 1)] : 0;
        }
        const int64_t q[4] = {P.off[f], P.ldf[f], P.woff[f], P.l21off[f]};
        std::memcpy(m + 8, q, sizeof q);
    }
    const size_t o_tmeta = put32(tmeta);
    // one 64-byte record per work item (kernels.h: CholPlanDev::fat); only the tile / row-tile items are
    // read through it, the others just keep the indexing simple
    std::vector<int32_t> fat(16 * (P.wl.size() / 3), 0);
    for (size_t i = 0; i < P.wl.size() / 3; i++)
    {
        int32_t* m = fat.data() + 16 * i;
        const int f = P.wl[3 * i];
        m[0] = f, m[1] = P.wl[3 * i + 1], m[2] = P.wl[3 * i + 2];
        if (f < 0 || f >= P.n_super)
            continue;
        m[3] = 6 * P.ncb[f], m[4] = 6 * (P.nb[f] - P.ncb[f]);
        m[5] = P.bw_np[f], m[6] = P.col0[f], m[7] = P.rows_ptr[f]; // (the ahead-of-time mat-vec items of the backward pass)
        const int64_t q[4] = {P.off[f], P.ldf[f], P.woff[f], P.l21off[f]};
        std::memcpy(m + 8, q, sizeof q);
        if ((int)i >= P.asm0 && (int)i < P.asm0 + P.nasm)
        { // assembly item: block rows of the front, where its map starts, and whether these block columns hold
          // anything but zeros (most of them are fill / update block: those need no map and no Hsc loads)
            const int64_t nb = P.nb[f], mo = P.asm_off[f];
            m[5] = (int32_t)nb;
            std::memcpy(m + 6, &mo, sizeof mo);
            bool any = false;
            for (int64_t cb = m[1]; cb < m[2]; cb++)
            {
                any = any || P.asm_map[mo + cb] >= 0;
                const int64_t c0 = mo + nb + cb * nb - cb * (cb - 1) / 2;
                for (int64_t k = 0; k < nb - cb && !any; k++)
                    any = P.asm_map[c0 + k] >= 0;
            }
            m[4] = any ? 1 : 0;
        }
    }
    const size_t o_fat = put32(fat);
    const size_t o_asm_map = put32(P.asm_map);
    const size_t o_trans = pack32.size(); // bytes
    pack32.resize(o_trans + (P.blk_trans.size() + 3) / 4 + 4, 0);
    if (!P.blk_trans.empty())
        std::memcpy(pack32.data() + o_trans, P.blk_trans.data(), P.blk_trans.size());
    const size_t o_asm_off = put64(P.asm_off);
    const size_t o_off = put64(P.off), o_woff = put64(P.woff), o_l21off = put64(P.l21off), o_ldf = put64(P.ldf);
    po = {o_ncb, o_nb, o_col0, o_alias, o_bwnp, o_lanp, o_rows_ptr, o_rows, o_child_ptr, o_child, o_rel_ptr, o_rel,
          o_task_ptr, o_task_fronts, o_blk_front, o_blk_row, o_blk_col, o_perm, o_col_front, o_wl, o_ea1, o_tmeta,
          o_fat, o_asm_map, o_trans, o_asm_off, o_off, o_woff, o_l21off, o_ldf};
}

void cugo_chol::upload(hipStream_t s)
{
    const CholPlan& P = plan;
    const size_t o_ncb = po[0], o_nb = po[1], o_col0 = po[2], o_alias = po[3], o_bwnp = po[4], o_lanp = po[5];
    const size_t o_rows_ptr = po[6], o_rows = po[7], o_child_ptr = po[8], o_child = po[9], o_rel_ptr = po[10];
    const size_t o_rel = po[11], o_task_ptr = po[12], o_task_fronts = po[13], o_blk_front = po[14], o_blk_row = po[15];
    const size_t o_blk_col = po[16], o_perm = po[17], o_col_front = po[18], o_wl = po[19], o_ea1 = po[20];
    const size_t o_tmeta = po[21], o_fat = po[22], o_asm_map = po[23], o_trans = po[24], o_asm_off = po[25];
    const size_t o_off = po[26], o_woff = po[27], o_l21off = po[28], o_ldf = po[29];
    d_pack32.upload(pack32, s), d_pack64.upload(pack64, s);
    d_fronts.resize((size_t)P.front_doubles + 16);
    d_fronts.zero(s); // once: afterwards only the lower triangles are cleared (k_clear_fronts)
    d_xnew.resize((size_t)6 * P.n + 16);
    d_junk.resize(64 * 1024);
    d_winv.resize((size_t)P.winv_doubles + 16);
    d_l21.resize((size_t)P.l21_doubles + 16);
    CUGO_HIP(hipStreamSynchronize(s)); // host vectors may be reused after return

    cugo_k::CholPlanDev& D = dev;
    D.n_fronts = P.n_super;
    const int32_t* b32 = d_pack32.data();
    const int64_t* b64 = d_pack64.data();
    D.ncb = b32 + o_ncb, D.nb = b32 + o_nb, D.off = b64 + o_off, D.col0 = b32 + o_col0;
    D.rows_ptr = b32 + o_rows_ptr, D.rows = b32 + o_rows;
    D.child_ptr = b32 + o_child_ptr, D.child = b32 + o_child;
    D.rel_ptr = b32 + o_rel_ptr, D.rel = b32 + o_rel;
    D.n_stages = P.n_stages;
    D.task_ptr = b32 + o_task_ptr, D.task_fronts = b32 + o_task_fronts, D.tmeta = b32 + o_tmeta;
    D.n_hsc_blocks = (int)P.blk_front.size();
    D.blk_front = b32 + o_blk_front, D.blk_row = b32 + o_blk_row;
    D.blk_col = b32 + o_blk_col, D.blk_trans = reinterpret_cast<const uint8_t*>(b32 + o_trans);
    D.n = P.n, D.perm = b32 + o_perm, D.col_front = b32 + o_col_front;
    D.junk = d_junk.data();
    D.woff = b64 + o_woff, D.winv = d_winv.data(), D.nc_max = P.nc_max;
    D.ea_lds = opt.ea_lds, D.panel16 = opt.panel16;
#ifdef CUGO_DEBUG_HOOKS
    {
        const char* ka = std::getenv("CUGO_KERNEL_ACQUIRE");
        D.kernel_acquire = ka ? std::atoi(ka) & 7 : 0; // 1: acquire at the start, 2: release at the end, 3: both, 4: waves wait for their stores
        const char* ed = std::getenv("CUGO_DEBUG_DELAY");
        D.dbg_delay = ed ? std::atoi(ed) : 0;
        const char* ez = std::getenv("CUGO_DEBUG_ZERO_LDS");
        D.zero_lds = ez ? std::atoi(ez) : 0, D.lds_doubles = 0;
        D.dbg_skip_wg = -1;
    }
#endif
    D.l21off = b64 + o_l21off, D.l21 = d_l21.data();
    D.ldf = b64 + o_ldf, D.alias_of = b32 + o_alias, D.bw_np = b32 + o_bwnp, D.la_np = b32 + o_lanp;
    d_wl_ptr = b32 + o_wl;
    D.wl_base = d_wl_ptr, D.fat = b32 + o_fat, D.ea1 = b32 + o_ea1;
    D.asm_map = b32 + o_asm_map, D.asm_off = b64 + o_asm_off;
    asm_fronts = opt.asm_fronts; // (CUGO_ASM_FRONTS=0: clear + scatter, two launches)
    lds_factor = cugo_k::chol_lds_factor_bytes(P.nc_max);
    lds_backward = cugo_k::chol_lds_backward_bytes(P.nc_max, P.ld_max);
}

void cugo_chol::analyze(int n, const int32_t* rowptr, const int32_t* colind)
{
    analyze_host(n, rowptr, colind);
    if (ctx)
        upload(ctx->stream);
}

// ordering + symbolic factorisation only (host); upload() brings the plan to the device
void cugo_chol::analyze_host(int n, const int32_t* rowptr, const int32_t* colind)
{
    CholOptions copt = CholOptions::from_env();
    // CUGO_OWN_SUBTREES=0: every rank factors everything (the replicated form of rounds 1-2)
    // unset: rank-owned subtrees only where the factorisation has work to divide.  On a graph whose levels all
    // run at the latency floor of their launches (kitti_00 shape: 1.6 GFLOP over 17 levels, every level 18-45 us
    // whatever its number of fronts) a rank that factors an eighth of a level's fronts finishes the level no
    // sooner, and the broadcasts at the ownership boundary come on top; CUGO_OWN_SUBTREES=1 forces the form —
    // also under a ONE-rank communicator (every subtree then belongs to rank 0, the top of the tree is
    // "replicated" on one rank, and every broadcast / reduce-scatter of the form is really issued: the
    // rehearsal of the multi-GPU path that a single GPU allows)
    const bool own_forced = opt.own_subtrees == 1;
    if ((world > 1 || own_forced) && bcast && opt.own_subtrees != 0)
        copt.rank = rank, copt.world = world, copt.owned = true;
    chol_analyze(n, rowptr, colind, copt, plan);
    if (copt.owned && !own_forced && plan.flops < 1e9 * opt.own_min_gflop)
    {
        copt.rank = 0, copt.world = 1, copt.owned = false;
        chol_analyze(n, rowptr, colind, copt, plan);
    }
    lookahead = opt.lookahead;
    trans32.assign(plan.blk_trans.begin(), plan.blk_trans.end());
    pack();
    if (plan.nc_max > cugo_k::chol_max_pivot_cols() ||
        cugo_k::chol_lds_factor_bytes(plan.nc_max) > 160 * 1024 ||
        cugo_k::chol_lds_backward_bytes(plan.nc_max, plan.ld_max) > 160 * 1024)
        throw std::runtime_error("cugo: a front exceeds the LDS budget (nc=" +
                                 std::to_string(plan.nc_max) + ", ld=" + std::to_string(plan.ld_max) + ")");
    analyzed = true;
}

#ifdef CUGO_DEBUG_HOOKS
// ---- diagnosis (libcugo_hip_hooks.so only) -------------------------------------------------
static cugo_chol* g_last_kept = nullptr; // the solver that ran last with CUGO_DEBUG_KEEP (cleared when it is destroyed)
static cugo_chol* g_ref_kept = nullptr;  // the reference run's solver, pinned by the autopsy tool
cugo_chol::~cugo_chol()
{
    if (g_last_kept == this)
        g_last_kept = nullptr;
    if (g_ref_kept == this)
        g_ref_kept = nullptr;
}
void cugo_chol::dump_kept(const char* dir)
{
    CUGO_HIP(hipStreamSynchronize(ctx->stream));
    const size_t nf = (size_t)plan.front_doubles, nw = (size_t)plan.winv_doubles, nl = (size_t)plan.l21_doubles,
                 nx = (size_t)6 * plan.n;
    for (size_t c = 0; c < keep.size(); c++)
    {
        std::vector<double> h(nf + nw + nl + 2 * nx);
        if (keep[c]->size() < h.size())
            continue;
        CUGO_HIP(hipMemcpy(h.data(), keep[c]->data(), h.size() * sizeof(double), hipMemcpyDeviceToHost));
        const std::string path = std::string(dir) + "/call" + std::to_string(c) + ".bin";
        std::FILE* f = std::fopen(path.c_str(), "wb");
        if (!f)
            throw std::runtime_error("cugo_debug_dump: cannot write " + path);
        const int64_t hdr[8] = {(int64_t)nf, (int64_t)nw, (int64_t)nl, (int64_t)nx, (int64_t)nx, 0, 0, 0};
        std::fwrite(hdr, sizeof hdr, 1, f);
        std::fwrite(h.data(), sizeof(double), h.size(), f);
        std::fclose(f);
    }
}
int cugo_debug_dump_last_solver(const char* dir)
{
    if (!g_last_kept)
        return 0;
    g_last_kept->dump_kept(dir);
    return (int)g_last_kept->keep.size();
}
cugo_chol* cugo_debug_solver(int which) { return which ? g_ref_kept : g_last_kept; }
void cugo_debug_pin_reference_solver() { g_ref_kept = g_last_kept; }
void cugo_chol::dump_slot(int call, const char* path)
{
    CUGO_HIP(hipStreamSynchronize(ctx->stream));
    const size_t nf = (size_t)plan.front_doubles, nw = (size_t)plan.winv_doubles, nl = (size_t)plan.l21_doubles,
                 nx = (size_t)6 * plan.n, total = nf + nw + nl + 2 * nx;
    if (call < 0 || call >= (int)keep.size() || keep[call]->size() < total)
        throw std::runtime_error("cugo_debug_dump_call: no such slot");
    std::vector<double> h(total);
    CUGO_HIP(hipMemcpy(h.data(), keep[call]->data(), total * sizeof(double), hipMemcpyDeviceToHost));
    std::FILE* f = std::fopen(path, "wb");
    if (!f)
        throw std::runtime_error(std::string("cugo_debug_dump_call: cannot write ") + path);
    const int64_t hdr[8] = {(int64_t)nf, (int64_t)nw, (int64_t)nl, (int64_t)nx, (int64_t)nx, 0, 0, 0};
    std::fwrite(hdr, sizeof hdr, 1, f);
    std::fwrite(h.data(), sizeof(double), h.size(), f);
    std::fclose(f);
}

// what factor_solve() consults between its launches in the hooks build: fault injection (CUGO_DEBUG_SKIP), stale-line
// injection (CUGO_DEBUG_STALE), in-stream checksums (CUGO_DEBUG_HASH, set by the engine), an empty kernel after every
// level (CUGO_DEBUG_GAP), copies of every factorisation's arrays (CUGO_DEBUG_KEEP)
struct cugo_chol_hooks
{
    cugo_chol& c;
    hipStream_t s;
    bool skip_on = false;
    double* stale_line = nullptr;
    int stale_stage = -1, stale_kind = -1;
    bool hash_levels, gap, keep_on;
    explicit cugo_chol_hooks(cugo_chol& solver) : c(solver), s(solver.ctx->stream)
    {
        const char* hl = std::getenv("CUGO_DEBUG_HASH_LEVELS");
        hash_levels = !(hl && hl[0] == '0'); // (0: only the checksums before and behind the loops — those between the
                                             // levels separate the launches, and the deviation they localise stays away)
        gap = std::getenv("CUGO_DEBUG_GAP") != nullptr;
        keep_on = std::getenv("CUGO_DEBUG_KEEP") != nullptr;
        const cugo_host::CholPlan& plan = c.plan;
        // CUGO_DEBUG_SKIP=call:launch:workgroup — in this solver's call number `call` the given workgroup of the given
        // launch returns at once; CUGO_DEBUG_SKIP_DUMP=file: call 0 writes its launch table
        const char* sk = std::getenv("CUGO_DEBUG_SKIP");
        const char* dump = c.dbg_calls == 0 ? std::getenv("CUGO_DEBUG_SKIP_DUMP") : nullptr;
        int cc = -1, l = -1, w = -1;
        if (sk && std::sscanf(sk, "%d:%d:%d", &cc, &l, &w) == 3 && cc == c.dbg_calls)
            cugo_k::chol_dbg_skip_begin(l, w, dump), skip_on = true;
        else if (dump)
            cugo_k::chol_dbg_skip_begin(-1, -1, dump), skip_on = true;
        c.dbg_calls++;
        // CUGO_DEBUG_STALE=call:kind:line — in call number `call` the 128-byte line `line` of W (kind 0) or of x in
        // elimination order (kind 1) shows the PREVIOUS factorisation's content to the one launch that follows the
        // launch that writes it and the right content to everything later: a write that becomes visible one launch late
        const char* sl = std::getenv("CUGO_DEBUG_STALE");
        int kind = -1;
        long line = -1;
        if (sl && std::sscanf(sl, "%d:%d:%ld", &cc, &kind, &line) == 3 && cc == c.dbg_calls - 1 && line >= 0)
        {
            std::vector<int> stage_of(plan.n_super, 0);
            for (int st = 0; st < plan.n_stages; st++)
                for (int t = plan.stage_task_ptr[st]; t < plan.stage_task_ptr[st + 1]; t++)
                    for (int fi = plan.task_ptr[t]; fi < plan.task_ptr[t + 1]; fi++)
                        stage_of[plan.task_fronts[fi]] = st;
            const int64_t e = 16 * (int64_t)line;
            if (kind == 0 && e + 16 <= plan.winv_doubles)
            {
                int f = 0;
                for (int k = 0; k < plan.n_super; k++)
                    if (plan.woff[k] <= e && plan.woff[k] >= plan.woff[f])
                        f = k;
                stale_line = c.d_winv.data() + e, stale_stage = stage_of[f], stale_kind = 0;
            }
            else if (kind == 1 && e + 16 <= 6LL * plan.n)
                stale_line = c.d_xnew.data() + e, stale_stage = stage_of[plan.col_front[e / 6]], stale_kind = 1;
            if (stale_line)
            {
                c.dbg_scratch.resize(16);
                CUGO_HIP(hipMemcpyAsync(c.dbg_scratch.data(), stale_line, 16 * sizeof(double), hipMemcpyDeviceToDevice, s));
            }
        }
    }
    ~cugo_chol_hooks()
    {
        if (skip_on)
            cugo_k::chol_dbg_skip_end();
    }
    void hash(int slot, const double* ptr, size_t n)
    {
        // 11 fronts after the assembly, 12 / 13 / 14 W, L21, fronts after the forward pass, 15 x after the backward
        // pass, 16 + st: W after the potrf launch of stage st, 40 + st: the fronts after its tile launches (a plan
        // with more than 24 stages keeps the first 24 of each kind: the two ranges must not overlap)
        if (c.dbg_hash && slot < 64)
            cugo_k::launch_hash_words(s, ptr, n, c.dbg_hash + slot);
    }
    void after_level(int st)
    {
        if (hash_levels && st < 24)
        {
            hash(16 + st, c.d_winv.data(), (size_t)c.plan.winv_doubles);
            hash(40 + st, c.d_fronts.data(), (size_t)c.plan.front_doubles);
        }
        if (gap)
            cugo_k::launch_nop(s);
    }
    double* stale_for_potrf(int st) const { return stale_kind == 0 && stale_stage == st ? stale_line : nullptr; }
    void after_two_phase(int st)
    {
        if (stale_kind == 0 && stale_stage == st)
            cugo_k::launch_swap16(s, stale_line, c.dbg_scratch.data());
    }
    void after_backward(int st)
    {
        if (stale_kind == 1 && (st == stale_stage || st == stale_stage - 1))
            cugo_k::launch_swap16(s, stale_line, c.dbg_scratch.data()); // (old content for the next level down only)
    }
    void keep_arrays(double* d_x)
    {
        if (!keep_on || c.dbg_calls > 16)
            return; // (dbg_calls counts this call already)
        const cugo_host::CholPlan& plan = c.plan;
        const size_t nf = (size_t)plan.front_doubles, nw = (size_t)plan.winv_doubles, nl = (size_t)plan.l21_doubles,
                     nx = (size_t)6 * plan.n;
        while (c.keep.size() < (size_t)c.dbg_calls)
            c.keep.emplace_back(new cugo_host::DevBuf<double>());
        auto& k = *c.keep[c.dbg_calls - 1];
        k.resize(nf + nw + nl + 2 * nx + 8);
        double* dst = k.data();
        const std::pair<const double*, size_t> parts[] = {{c.d_fronts.data(), nf}, {c.d_winv.data(), nw}, {c.d_l21.data(), nl},
                                                          {c.d_xnew.data(), nx}, {d_x, nx}};
        for (const auto& pr : parts)
        {
            if (pr.second)
                CUGO_HIP(hipMemcpyAsync(dst, pr.first, pr.second * sizeof(double), hipMemcpyDeviceToDevice, s));
            dst += pr.second;
        }
        g_last_kept = &c;
    }
};
#define CUGO_HOOK(expr) expr
#else
cugo_chol::~cugo_chol() {}
#define CUGO_HOOK(expr) ((void)0)
#endif

void cugo_chol::factor_solve(const double* d_Hsc, double lambda, const double* d_bsc, double* d_x,
                             int32_t* d_fail)
{
    hipStream_t s = ctx->stream;
#ifdef CUGO_DEBUG_HOOKS
    cugo_chol_hooks hooks(*this);
#endif
#ifdef CUGO_STAMPS // (make STAMPS=1: in-kernel cycle stamps of the last launch of every kernel, printed by the fifth call)
    static const bool dbg = std::getenv("CUGO_DEBUG_STAMPS") != nullptr;
    static long long* d_stamps = nullptr;
    if (dbg && !d_stamps)
    {
        CUGO_HIP(hipMalloc(reinterpret_cast<void**>(&d_stamps), 64 * sizeof(long long)));
        CUGO_HIP(hipMemset(d_stamps, 0, 64 * sizeof(long long)));
        cugo_k::set_debug_stamps(d_stamps);
    }
#endif
    cugo_k::launch_chol_assemble(s, dev, d_fronts.data(), (size_t)plan.front_doubles, d_Hsc, lambda, d_bsc, d_fail,
                                 d_wl_ptr + 3L * plan.clr0, plan.nclr, asm_fronts ? d_wl_ptr + 3L * plan.asm0 : nullptr,
                                 plan.nasm);
    CUGO_HOOK(hooks.hash(11, d_fronts.data(), (size_t)plan.front_doubles));
    int pend0 = 0, npend = 0, pend_tile = 64; // update tiles of the previous level, not launched yet
    for (int st = 0; st < plan.n_stages; st++)
    {
        const int t0 = plan.stage_task_ptr[st], t1 = plan.stage_task_ptr[st + 1];
        if (plan.has_subtree_stage && st == 0)
            cugo_k::launch_chol_subtree_stage(s, dev, d_fronts.data(), t0, t1 - t0, lds_factor, d_fail);
        else if (lookahead)
        {
            cugo_k::launch_chol_potrf_la(s, dev, d_fronts.data(), t0, t1 - t0, d_wl_ptr + 3L * pend0, npend,
                                         pend_tile, d_fail);
            cugo_k::launch_chol_lead(s, dev, d_fronts.data(), d_wl_ptr + 3L * plan.lead_ptr[st],
                                     plan.lead_ptr[st + 1] - plan.lead_ptr[st], d_wl_ptr + 3L * plan.ea_ptr[st],
                                     plan.ea_ptr[st + 1] - plan.ea_ptr[st], d_wl_ptr + 3L * plan.eab_ptr[st],
                                     plan.eab_ptr[st + 1] - plan.eab_ptr[st]);
            pend0 = plan.sb_ptr[st], npend = plan.sb_ptr[st + 1] - plan.sb_ptr[st];
            pend_tile = plan.stage_tile[st] ? plan.stage_tile[st] : 64; // (its item lists are 64x64 ones)
        }
        else
        {
            double* stale = nullptr;
            double* stale_scratch = nullptr;
            CUGO_HOOK((stale = hooks.stale_for_potrf(st), stale_scratch = dbg_scratch.data()));
            cugo_k::launch_chol_upper_stage(
                s, dev, d_fronts.data(), t0, t1 - t0, d_wl_ptr, plan.ea_ptr[st],
                plan.ea_ptr[st + 1] - plan.ea_ptr[st], plan.eab_ptr[st],
                plan.eab_ptr[st + 1] - plan.eab_ptr[st], plan.syrk_ptr[st],
                plan.syrk_ptr[st + 1] - plan.syrk_ptr[st], plan.stage_tile[st], lds_factor, d_fail, stale, stale_scratch);
            if (plan.stage_tile[st] == 0)
            {
                cugo_k::launch_chol_two_phase(s, dev, d_fronts.data(), d_wl_ptr + 3L * plan.trsm_ptr[st],
                                              plan.trsm_ptr[st + 1] - plan.trsm_ptr[st],
                                              d_wl_ptr + 3L * plan.syrk_ptr[st],
                                              plan.syrk_ptr[st + 1] - plan.syrk_ptr[st]);
                CUGO_HOOK(hooks.after_two_phase(st));
            }
        }
        CUGO_HOOK(hooks.after_level(st));
        // update blocks that cross the ownership boundary: the subtree roots of this level whose parent is
        // replicated go from their owner to every rank (columns 6 ncb .. of the front: one contiguous range)
        bool grouped = false;
        for (size_t k = 0; k < plan.xu_front.size(); k++)
            if (plan.xu_stage[k] == st)
            {
                if (!grouped && bcast_group)
                    bcast_group(true), grouped = true;
                const int f = plan.xu_front[k];
                // from element (c0, c0) to the rhs-row entry of the last column: one contiguous range that stays
                // inside the front's storage also when the front lives in its child's update block
                const int64_t ld = plan.ldf[f], c0 = 6LL * plan.ncb[f], c1 = 6LL * plan.nb[f];
                if (c1 > c0)
                    bcast(d_fronts.data() + plan.off[f] + c0 * ld + c0, (size_t)((c1 - 1 - c0) * ld + c1 + 1 - c0),
                          plan.xu_owner[k]);
            }
        if (grouped)
            bcast_group(false);
    }
    if (npend > 0) // the last level's tiles (the rhs rows of the roots)
        cugo_k::launch_chol_potrf_la(s, dev, d_fronts.data(), 0, 0, d_wl_ptr + 3L * pend0, npend, pend_tile, d_fail);
    CUGO_HOOK(hooks.hash(12, d_winv.data(), (size_t)plan.winv_doubles));
    CUGO_HOOK(hooks.hash(13, d_l21.data(), (size_t)plan.l21_doubles));
    CUGO_HOOK(hooks.hash(14, d_fronts.data(), (size_t)plan.front_doubles));
    for (int st = plan.n_stages - 1; st >= 0; st--)
    {
        const int t0 = plan.stage_task_ptr[st], t1 = plan.stage_task_ptr[st + 1];
        // the level's fronts, plus the ahead-of-time mat-vecs of their children as extra workgroups
        cugo_k::launch_chol_backward_stage(s, dev, d_fronts.data(), t0, t1 - t0, lds_backward, d_xnew.data(),
                                           d_x, d_wl_ptr + 3L * plan.bwg_ptr[st],
                                           plan.bwg_ptr[st + 1] - plan.bwg_ptr[st]);
        CUGO_HOOK(hooks.after_backward(st));
#ifdef CUGO_STAMPS
        if (dbg && st == plan.n_stages - 1)
        { // keep the top stage's backward stamps (kernel 3) in slots 48.. before stage 0 overwrites them
            CUGO_HIP(hipStreamSynchronize(s));
            CUGO_HIP(hipMemcpy(d_stamps + 48, d_stamps + 24, 8 * sizeof(long long), hipMemcpyDeviceToDevice));
        }
#endif
    }
    CUGO_HOOK(hooks.hash(15, d_xnew.data(), (size_t)6 * plan.n));
    if (!plan.xx_lo.empty())
    { // the solution of the other ranks' subtrees, then the un-permutation of the whole vector
        if (bcast_group)
            bcast_group(true);
        for (size_t k = 0; k < plan.xx_lo.size(); k++)
            bcast(d_xnew.data() + 6LL * plan.xx_lo[k], (size_t)(6LL * (plan.xx_hi[k] - plan.xx_lo[k])), plan.xx_owner[k]);
        if (bcast_group)
            bcast_group(false);
        cugo_k::launch_chol_unpermute(s, dev, d_xnew.data(), d_x);
    }
    CUGO_HIP(hipGetLastError());
    CUGO_HOOK(hooks.keep_arrays(d_x));
#ifdef CUGO_STAMPS
    if (dbg)
    {
        long long h[64];
        CUGO_HIP(hipStreamSynchronize(s));
        CUGO_HIP(hipMemcpy(h, d_stamps, sizeof h, hipMemcpyDeviceToHost));
        static int calls = 0;
        if (++calls == 5)
            for (int k = 0; k < 7; k++)
            {
                std::printf("stamps kernel %d (nc=%lld):", k, h[k * 8 + 6]);
                for (int i = 1; i < 8; i++)
                    if (i != 6 && h[k * 8 + i])
                        std::printf(" [%d] +%lld", i, h[k * 8 + i] - h[k * 8]);
                std::printf("  (cycles since kernel start; last launch of the kernel)\n");
            }
    }
#endif
}
