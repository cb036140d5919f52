// Ordering (level-set nested dissection + minimum degree leaves), elimination tree, relaxed
// supernodes, front layout, extend-add maps and the stage/task schedule of the multifrontal
// LL^T.  Everything here works on the BLOCK graph (one node per free pose), so even the
// 10k-pose configuration is a 10k-node problem on the host.
//
// The reference delegates all of this to closed-source cuSOLVER/METIS on the scalar 6Px6P
// pattern (ref: src/cholesky.hpp:97-98,295-296; src/cuda_linear_solver.cpp:27-42).
#include "chol_symbolic.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <thread>

namespace cugo_host
{

CholOptions CholOptions::from_env()
{
    CholOptions o;
    if (const char* s = std::getenv("CUGO_ND_LEAF"))
        o.nd_leaf = std::max(1, std::atoi(s));
    if (const char* s = std::getenv("CUGO_MAX_SUPER_COLS"))
        o.max_super_cols = std::max(1, std::atoi(s));
    if (const char* s = std::getenv("CUGO_ZERO_FRAC"))
        o.zero_frac = std::atof(s);
    if (const char* s = std::getenv("CUGO_TARGET_TASKS"))
        o.target_tasks = std::max(1, std::atoi(s));
    if (const char* s = std::getenv("CUGO_ALIAS_CHAINS"))
        o.alias_chains = std::atoi(s) != 0;
    if (const char* s = std::getenv("CUGO_MIN_SUBTREE_TASKS"))
        o.min_subtree_tasks = std::max(0, std::atoi(s));
    if (const char* s = std::getenv("CUGO_MAX_FRONT_COLS"))
        o.max_front_cols = std::min(16, std::max(1, std::atoi(s)));
    if (const char* s = std::getenv("CUGO_TWO_PHASE_MIN_TILES"))
        o.two_phase_min_tiles = std::max(0, std::atoi(s));
    if (const char* s = std::getenv("CUGO_XCD_AFFINITY"))
        o.xcd_affinity = std::atoi(s) != 0;
    if (const char* s = std::getenv("CUGO_TILE32_MAX_TILES"))
        o.tile32_max_tiles = std::max(0, std::atoi(s));
    return o;
}

namespace
{

struct Graph
{
    int n = 0;
    std::vector<int> ptr, adj; // symmetric, no self loops, sorted
};

Graph build_graph(int n, const int32_t* rowptr, const int32_t* colind)
{
    Graph g;
    g.n = n;
    std::vector<int> deg(n + 1, 0);
    for (int r = 0; r < n; r++)
        for (int k = rowptr[r]; k < rowptr[r + 1]; k++)
            if (colind[k] != r)
            {
                deg[r + 1]++;
                deg[colind[k] + 1]++;
            }
    g.ptr.assign(n + 1, 0);
    for (int i = 0; i < n; i++)
        g.ptr[i + 1] = g.ptr[i] + deg[i + 1];
    g.adj.resize(g.ptr[n]);
    std::vector<int> pos(g.ptr.begin(), g.ptr.end() - 1);
    for (int r = 0; r < n; r++)
        for (int k = rowptr[r]; k < rowptr[r + 1]; k++)
            if (colind[k] != r)
            {
                g.adj[pos[r]++] = colind[k];
                g.adj[pos[colind[k]]++] = r;
            }
    for (int i = 0; i < n; i++)
    {
        std::sort(g.adj.begin() + g.ptr[i], g.adj.begin() + g.ptr[i + 1]);
    }
    return g;
}

// ------------------------------------------------------------------ nested dissection ---
struct ND
{
    const Graph& g;
    const CholOptions& opt;
    // setid / level are indexed by node: the two halves of a dissection own disjoint nodes, so they
    // can be ordered by different threads (the top levels are: see run()).  A thread only ever
    // compares the setid of a foreign node with its own id, hence the relaxed atomic accesses.
    // (one 8-byte record per node: the inner test of every BFS — same set? not visited yet? — is one cache line)
    struct NodeState
    {
        int sid, level;
    };
    std::vector<NodeState> st;
    std::atomic<int> next_id{1};
    ND(const Graph& g_, const CholOptions& o) : g(g_), opt(o), st(g_.n, NodeState{0, -1}) {}
    int sid(int v) const { return __atomic_load_n(&st[v].sid, __ATOMIC_RELAXED); }
    void set_sid(int v, int id) { __atomic_store_n(&st[v].sid, id, __ATOMIC_RELAXED); }

    // exact minimum degree on the subgraph induced by `nodes` (small sets only)
    void leaf_order(const std::vector<int>& nodes, std::vector<int>& order)
    {
        const int m = (int)nodes.size();
        if (m > 512)
        {
            for (int v : nodes)
                order.push_back(v);
            return;
        }
        const int id = next_id.fetch_add(1);
        std::vector<int> local(m);
        for (int i = 0; i < m; i++)
            set_sid(nodes[i], id);
        // local adjacency
        std::vector<std::vector<int>> adj(m);
        {
            std::vector<int> gl2l; // map through binary search on sorted copy
            std::vector<int> sorted(nodes);
            std::sort(sorted.begin(), sorted.end());
            std::vector<int> where(m);
            for (int i = 0; i < m; i++)
                where[std::lower_bound(sorted.begin(), sorted.end(), nodes[i]) - sorted.begin()] = i;
            for (int i = 0; i < m; i++)
                for (int k = g.ptr[nodes[i]]; k < g.ptr[nodes[i] + 1]; k++)
                {
                    const int u = g.adj[k];
                    if (sid(u) == id)
                        adj[i].push_back(
                            where[std::lower_bound(sorted.begin(), sorted.end(), u) - sorted.begin()]);
                }
        }
        // exact minimum degree (ties: lowest local index) on a bit matrix: eliminating `best` ORs its
        // row into the rows of its neighbours — O(m) words per step instead of O(deg^2) list edits
        const int W = (m + 63) >> 6;
        std::vector<uint64_t> bits((size_t)m * W, 0);
        for (int i = 0; i < m; i++)
            for (int u : adj[i])
                bits[(size_t)i * W + (u >> 6)] |= 1ull << (u & 63);
        std::vector<int> deg(m);
        for (int i = 0; i < m; i++)
            deg[i] = (int)adj[i].size();
        std::vector<char> done(m, 0);
        for (int step = 0; step < m; step++)
        {
            int best = -1;
            for (int i = 0; i < m; i++)
                if (!done[i] && (best < 0 || deg[i] < deg[best]))
                    best = i;
            done[best] = 1;
            order.push_back(nodes[best]);
            const uint64_t* rb = &bits[(size_t)best * W];
            for (int w = 0; w < W; w++)
            {
                uint64_t nb = rb[w];
                while (nb)
                {
                    const int u = (w << 6) + __builtin_ctzll(nb);
                    nb &= nb - 1;
                    uint64_t* ru = &bits[(size_t)u * W];
                    int d = 0;
                    for (int q = 0; q < W; q++)
                    {
                        ru[q] |= rb[q];
                        if (q == (u >> 6))
                            ru[q] &= ~(1ull << (u & 63));       // no self loop
                        if (q == (best >> 6))
                            ru[q] &= ~(1ull << (best & 63));    // best leaves the graph
                        d += __builtin_popcountll(ru[q]);
                    }
                    deg[u] = d;
                }
            }
        }
    }

    // BFS inside the set `id` from `src`; fills level[], returns visit order
    void bfs(int src, int id, std::vector<int>& out)
    {
        out.clear();
        out.push_back(src);
        st[src].level = 0;
        for (size_t h = 0; h < out.size(); h++)
        {
            const int v = out[h];
            for (int k = g.ptr[v]; k < g.ptr[v + 1]; k++)
            {
                const int u = g.adj[k];
                if (sid(u) == id && st[u].level < 0)
                {
                    st[u].level = st[v].level + 1;
                    out.push_back(u);
                }
            }
        }
    }

    // appends the ordering of `nodes` to `order`; depth < kParallelDepth: the two halves run on
    // two threads (the result does not depend on it: A's order, then B's, then the separator)
    const int kParallelDepth = std::getenv("CUGO_ND_PAR") ? std::atoi(std::getenv("CUGO_ND_PAR")) : 4; // 0: one thread
    void run(std::vector<int> nodes, std::vector<int>& order, int depth = 0)
    {
        if ((int)nodes.size() <= opt.nd_leaf)
        {
            leaf_order(nodes, order);
            return;
        }
        const int id = next_id.fetch_add(1);
        for (int v : nodes)
        {
            set_sid(v, id);
            st[v].level = -1;
        }
        // connected components
        std::vector<int> comp;
        bfs(nodes[0], id, comp);
        if (comp.size() < nodes.size())
        {
            std::vector<std::vector<int>> comps;
            comps.push_back(comp);
            for (int v : nodes)
                if (st[v].level < 0)
                {
                    bfs(v, id, comp);
                    comps.push_back(comp);
                }
            for (auto& c : comps)
                run(c, order, depth + 1);
            return;
        }
        // pseudo-peripheral start: repeat BFS from the last-visited node
        int src = comp.back();
        for (int it = 0; it < 3; it++)
        {
            for (int v : nodes)
                st[v].level = -1;
            bfs(src, id, comp);
            const int far = comp.back();
            if (it == 2 || st[far].level <= 1)
                break;
            int depth_before = st[far].level;
            (void)depth_before;
            src = far;
        }
        const int h = st[comp.back()].level;
        if (h < 2)
        {
            leaf_order(nodes, order);
            return;
        }
        std::vector<int> cnt(h + 1, 0);
        for (int v : nodes)
            cnt[st[v].level]++;
        // candidate separators: levels leaving >= 30% of the nodes on each side; the smallest wins
        const int total = (int)nodes.size();
        int best = -1;
        {
            int below = 0;
            double best_score = 1e300;
            for (int m = 0; m <= h; m++)
            {
                const int above = total - below - cnt[m];
                if (m >= 1 && m <= h - 1)
                {
                    const int mn = std::min(below, above);
                    const double bal = (double)mn / total;
                    double score = cnt[m] + (bal >= 0.3 ? 0.0 : 1e6 * (0.3 - bal));
                    if (score < best_score)
                    {
                        best_score = score;
                        best = m;
                    }
                }
                below += cnt[m];
            }
        }
        std::vector<int> A, B, S;
        for (int v : comp)
        {
            if (st[v].level < best)
                A.push_back(v);
            else if (st[v].level > best)
                B.push_back(v);
            else
            {
                // thin the separator: a level node without a neighbour above joins A
                bool touches_B = false;
                for (int k = g.ptr[v]; k < g.ptr[v + 1] && !touches_B; k++)
                {
                    const int u = g.adj[k];
                    if (sid(u) == id && st[u].level == best + 1)
                        touches_B = true;
                }
                (touches_B ? S : A).push_back(v);
            }
        }
        // (a thread costs ~30 us to start; a half of 150+ nodes of a graph this dense — ~50 neighbours per pose —
        // takes longer than that to dissect: the kitti_00 shape, 1 322 poses, orders in 0.5 instead of 1.2 ms)
        if (depth < kParallelDepth && A.size() + B.size() > 300)
        {
            std::vector<int> orderB;
            std::exception_ptr err;
            std::thread tb([&] {
                try
                {
                    run(std::move(B), orderB, depth + 1);
                }
                catch (...)
                {
                    err = std::current_exception();
                }
            });
            run(std::move(A), order, depth + 1);
            tb.join();
            if (err)
                std::rethrow_exception(err);
            order.insert(order.end(), orderB.begin(), orderB.end());
        }
        else
        {
            run(std::move(A), order, depth + 1);
            run(std::move(B), order, depth + 1);
        }
        for (int v : S)
            order.push_back(v);
    }
};

} // namespace

void chol_analyze(int n, const int32_t* rowptr, const int32_t* colind, const CholOptions& opt,
                  CholPlan& P)
{
    // CUGO_INIT_TIMING=1: per-phase host times on stderr (diagnosis only)
    const bool timing = std::getenv("CUGO_INIT_TIMING") != nullptr;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing)
            return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[cugo symbolic] %-26s %8.3f ms\n", what,
                     std::chrono::duration<double, std::milli>(now - lap_t).count());
        lap_t = now;
    };
    P = CholPlan();
    P.n = n;
    if (n == 0)
        return;
    const Graph g = build_graph(n, rowptr, colind);
    lap("0 adjacency graph");

    // ---- 1. fill-reducing, parallelism-exposing ordering --------------------------------
    std::vector<int> perm;
    {
        ND nd(g, opt);
        std::vector<int> all(n);
        std::iota(all.begin(), all.end(), 0);
        perm.reserve(n);
        nd.run(std::move(all), perm);
        if ((int)perm.size() != n)
            throw std::runtime_error("cugo: ordering lost nodes");
    }
    std::vector<int> iperm(n);
    for (int i = 0; i < n; i++)
        iperm[perm[i]] = i;

    lap("1 ordering");
    // ---- 2. elimination tree (Liu, path compression) + postorder -----------------------
    auto etree = [&](const std::vector<int>& ip, const std::vector<int>& pm, std::vector<int>& parent) {
        parent.assign(n, -1);
        std::vector<int> anc(n, -1);
        for (int j = 0; j < n; j++)
        {
            const int vj = pm[j];
            for (int k = g.ptr[vj]; k < g.ptr[vj + 1]; k++)
            {
                int i = ip[g.adj[k]];
                while (i != -1 && i < j)
                {
                    const int nx = anc[i];
                    anc[i] = j;
                    if (nx == -1)
                        parent[i] = j;
                    i = nx;
                }
            }
        }
    };
    std::vector<int> parent;
    etree(iperm, perm, parent);
    {
        // children lists (ascending), iterative DFS postorder
        std::vector<int> head(n, -1), next(n, -1);
        for (int j = n - 1; j >= 0; j--)
            if (parent[j] >= 0)
            {
                next[j] = head[parent[j]];
                head[parent[j]] = j;
            }
        std::vector<int> post;
        post.reserve(n);
        std::vector<int> stack;
        for (int r = 0; r < n; r++)
        {
            if (parent[r] != -1)
                continue;
            stack.push_back(r);
            while (!stack.empty())
            {
                const int v = stack.back();
                const int c = head[v];
                if (c == -1)
                {
                    post.push_back(v);
                    stack.pop_back();
                }
                else
                {
                    head[v] = next[c];
                    stack.push_back(c);
                }
            }
        }
        std::vector<int> perm2(n);
        for (int k = 0; k < n; k++)
            perm2[k] = perm[post[k]];
        perm.swap(perm2);
        for (int i = 0; i < n; i++)
            iperm[perm[i]] = i;
        etree(iperm, perm, parent);
    }

    lap("2 etree + postorder");
    // ---- 3. column structures of L (block level) ---------------------------------------
    std::vector<int> cptr(n + 1, 0);
    std::vector<int> cidx; // struct(j): rows > j, sorted
    std::vector<int> nchild(n, 0);
    {
        std::vector<int> head(n, -1), next(n, -1);
        for (int j = n - 1; j >= 0; j--)
            if (parent[j] >= 0)
            {
                next[j] = head[parent[j]];
                head[parent[j]] = j;
                nchild[parent[j]]++;
            }
        std::vector<int> mark(n, -1), tmp;
        cidx.reserve((size_t)n * 16);
        // Up to 2 048 block columns the structures are formed as bit rows (n / 64 <= 32 words per column): struct(j) is
        // the OR of the children's rows and of j's own neighbours, and the set bits come out in ascending order — no
        // marker pass over every child entry and no sort per column (the kitti_00 shape, 1 322 columns: 2 - 3 x faster,
        // on the helper thread's chain of a new-graph call).  Larger graphs keep the list form below: their rows get
        // long compared with their structures (10 000 columns: 157 words per row for ~100 entries — measured 5 x SLOWER).
        const bool bit_rows = n <= 2048;
        const int W = (n + 63) >> 6;
        std::vector<uint64_t> rowbits(bit_rows ? (size_t)n * W : 0, 0);
        for (int j = 0; bit_rows && j < n; j++)
        {
            uint64_t* rj = &rowbits[(size_t)j * W];
            const int vj = perm[j];
            for (int k = g.ptr[vj]; k < g.ptr[vj + 1]; k++)
            {
                const int i = iperm[g.adj[k]];
                if (i > j)
                    rj[i >> 6] |= 1ull << (i & 63);
            }
            const int w0 = j >> 6; // (nothing at or below j survives: children only hold rows > themselves, j itself is cleared)
            for (int c = head[j]; c != -1; c = next[c])
            {
                const uint64_t* rc = &rowbits[(size_t)c * W];
                for (int w = w0; w < W; w++)
                    rj[w] |= rc[w];
            }
            rj[w0] &= ~((j & 63) == 63 ? ~0ull : ((2ull << (j & 63)) - 1)); // bits <= j of the word that holds j
            for (int w = w0; w < W; w++)
            {
                uint64_t b = rj[w];
                while (b)
                {
                    cidx.push_back((w << 6) + __builtin_ctzll(b));
                    b &= b - 1;
                }
            }
            cptr[j + 1] = (int)cidx.size();
        }
        for (int j = 0; !bit_rows && j < n; j++)
        {
            tmp.clear();
            mark[j] = j;
            const int vj = perm[j];
            for (int k = g.ptr[vj]; k < g.ptr[vj + 1]; k++)
            {
                const int i = iperm[g.adj[k]];
                if (i > j && mark[i] != j)
                {
                    mark[i] = j;
                    tmp.push_back(i);
                }
            }
            for (int c = head[j]; c != -1; c = next[c])
                for (int k = cptr[c]; k < cptr[c + 1]; k++)
                {
                    const int i = cidx[k];
                    if (i > j && mark[i] != j)
                    {
                        mark[i] = j;
                        tmp.push_back(i);
                    }
                }
            std::sort(tmp.begin(), tmp.end());
            cidx.insert(cidx.end(), tmp.begin(), tmp.end());
            cptr[j + 1] = (int)cidx.size();
        }
    }
    auto csize = [&](int j) { return cptr[j + 1] - cptr[j]; };

    lap("3 column structures");
    // ---- 4. fundamental supernodes, then relaxed amalgamation --------------------------
    std::vector<int> sfirst; // first column of each supernode (ascending)
    for (int j = 0; j < n; j++)
    {
        const bool cont = j > 0 && parent[j - 1] == j && nchild[j] == 1 && csize(j - 1) == csize(j) + 1;
        if (!cont)
            sfirst.push_back(j);
    }
    int ns = (int)sfirst.size();
    sfirst.push_back(n);
    {
        // s -> [first,last]; parent supernode via parent[last]
        std::vector<int> first(sfirst.begin(), sfirst.end() - 1), last(ns), alive(ns, 1);
        for (int s = 0; s < ns; s++)
            last[s] = sfirst[s + 1] - 1;
        std::vector<int> col2s(n);
        for (int s = 0; s < ns; s++)
            for (int j = first[s]; j <= last[s]; j++)
                col2s[j] = s;
        // zeros[s]: explicit zero blocks already accepted in supernode s
        std::vector<double> zeros(ns, 0.0);
        for (int p = 0; p < ns; p++)
        {
            if (!alive[p])
                continue;
            for (;;)
            {
                const int cl = first[p] - 1; // candidate child = supernode ending right before p
                if (cl < 0)
                    break;
                const int c = col2s[cl];
                if (!alive[c] || parent[cl] != first[p])
                    break;
                const int nc_c = last[c] - first[c] + 1, nc_p = last[p] - first[p] + 1;
                if (nc_c + nc_p > opt.max_super_cols)
                    break;
                const int rows_p = csize(last[p]);           // boundary of p
                const int rows_c = csize(last[c]);           // boundary of c (includes p's cols it hits)
                const double extra = (double)nc_c * (nc_p + rows_p - rows_c);
                const double tot_blocks = (double)(nc_c + nc_p) * (nc_c + nc_p + 1) / 2 +
                                          (double)(nc_c + nc_p) * rows_p;
                const double z = zeros[p] + zeros[c] + extra;
                if (nc_c + nc_p > 2 && z > opt.zero_frac * tot_blocks)
                    break;
                // merge c into p
                for (int j = first[c]; j <= last[c]; j++)
                    col2s[j] = p;
                first[p] = first[c];
                zeros[p] = z;
                alive[c] = 0;
            }
        }
        std::vector<int> nf;
        for (int s = 0; s < ns; s++)
            if (alive[s])
                nf.push_back(first[s]);
        std::sort(nf.begin(), nf.end());
        nf.push_back(n);
        // split supernodes wider than the LDS cap into a chain of panels (each panel's
        // boundary is the rest of the supernode plus the original boundary: still nested)
        sfirst.clear();
        for (size_t i = 0; i + 1 < nf.size(); i++)
        {
            const int a = nf[i], b = nf[i + 1];
            const int w = b - a;
            const int parts = (w + opt.max_front_cols - 1) / opt.max_front_cols;
            for (int q = 0; q < parts; q++)
                sfirst.push_back(a + (int)((int64_t)w * q / parts));
        }
        ns = (int)sfirst.size();
        sfirst.push_back(n);
    }

    P.perm.assign(perm.begin(), perm.end());
    P.iperm.assign(iperm.begin(), iperm.end());
    P.n_super = ns;
    P.super_ptr.assign(sfirst.begin(), sfirst.end());
    P.col_front.resize(n);
    for (int s = 0; s < ns; s++)
        for (int j = sfirst[s]; j < sfirst[s + 1]; j++)
            P.col_front[j] = s;

    lap("4 supernodes");
    // ---- 5. fronts: boundary rows, parents, relative indices ---------------------------
    P.rows_ptr.assign(ns + 1, 0);
    P.ncb.resize(ns), P.nb.resize(ns), P.col0.resize(ns), P.off.resize(ns), P.sparent.assign(ns, -1);
    P.woff.resize(ns);
    int64_t off = 0, woff = 0;
    for (int s = 0; s < ns; s++)
    {
        const int lastc = sfirst[s + 1] - 1;
        const int nr = csize(lastc);
        P.rows_ptr[s + 1] = P.rows_ptr[s] + nr;
        P.rows.insert(P.rows.end(), cidx.begin() + cptr[lastc], cidx.begin() + cptr[lastc + 1]);
        P.ncb[s] = sfirst[s + 1] - sfirst[s];
        P.nb[s] = P.ncb[s] + nr;
        P.col0[s] = sfirst[s];
        P.off[s] = off;
        const int64_t ld = 6LL * P.nb[s] + 1;
        off += ld * 6LL * P.nb[s];
        const int64_t ncp = (6LL * P.ncb[s] + 15) & ~15LL;
        P.woff[s] = woff;
        woff += ncp * ncp;
        P.ld_max = std::max<long>(P.ld_max, (long)ld);
        if (nr > 0)
            P.sparent[s] = P.col_front[P.rows[P.rows_ptr[s]]];
        const double nc = 6.0 * P.ncb[s], nrs = 6.0 * nr;
        P.nnzL += nc * (nc + 1) / 2 + nc * nrs;
        P.flops += 2.0 * (nc * nc * nc / 6.0 + nc * nc * nrs / 2.0 + nc * nrs * nrs / 2.0) +
                   nc * nc * nrs; // LL^T of the pivot block + TRSM + SYRK of the update
    }
    P.front_doubles = off;
    P.winv_doubles = woff;
    // children (ascending order => fixed, deterministic extend-add order)
    P.child_ptr.assign(ns + 1, 0);
    for (int s = 0; s < ns; s++)
        if (P.sparent[s] >= 0)
            P.child_ptr[P.sparent[s] + 1]++;
    for (int s = 0; s < ns; s++)
        P.child_ptr[s + 1] += P.child_ptr[s];
    P.child.resize(P.child_ptr[ns]);
    {
        std::vector<int> pos(P.child_ptr.begin(), P.child_ptr.end() - 1);
        for (int s = 0; s < ns; s++)
            if (P.sparent[s] >= 0)
                P.child[pos[P.sparent[s]]++] = s;
    }
    P.rel_ptr.assign(ns + 1, 0);
    for (int s = 0; s < ns; s++)
        P.rel_ptr[s + 1] = P.rel_ptr[s] + (P.rows_ptr[s + 1] - P.rows_ptr[s]);
    P.rel.resize(P.rel_ptr[ns]);
    for (int s = 0; s < ns; s++)
    {
        const int p = P.sparent[s];
        if (p < 0)
            continue;
        const int pf = sfirst[p], pl = sfirst[p + 1]; // parent's columns [pf,pl)
        const int* prow = P.rows.data() + P.rows_ptr[p];
        const int pnr = P.rows_ptr[p + 1] - P.rows_ptr[p];
        int q = 0;
        for (int k = P.rows_ptr[s]; k < P.rows_ptr[s + 1]; k++)
        {
            const int r = P.rows[k];
            int pos;
            if (r < pl)
            {
                if (r < pf)
                    throw std::runtime_error("cugo: symbolic: child row below parent front");
                pos = r - pf;
            }
            else
            {
                while (q < pnr && prow[q] < r)
                    q++;
                if (q >= pnr || prow[q] != r)
                    throw std::runtime_error("cugo: symbolic: child row missing in parent front");
                pos = P.ncb[p] + q;
            }
            P.rel[P.rel_ptr[s] + (k - P.rows_ptr[s])] = pos;
        }
    }

    lap("5 fronts");
    // ---- 5b. storage: single-child chains share memory --------------------------------------
    // A front whose only child has exactly the front's rows as its boundary (the pieces of a wide
    // supernode, and most single-child links of the etree) would receive that child's update
    // block by a plain copy.  It is stored IN the child's update block instead (same leading
    // dimension, rhs row included): no extend-add for it, and the memory is not allocated twice.
    P.ldf.assign(ns, 0), P.alias_of.assign(ns, -1);
    P.n_aliased = 0;
    {
        int64_t o = 0;
        const bool enable = opt.alias_chains;
        for (int s = 0; s < ns; s++) // postorder: children first
        {
            int c = -1;
            if (enable && P.child_ptr[s + 1] - P.child_ptr[s] == 1)
            {
                c = P.child[P.child_ptr[s]];
                const int nbr = P.nb[c] - P.ncb[c];
                bool same = nbr == P.nb[s];
                for (int i = 0; same && i < nbr; i++)
                    same = P.rel[P.rel_ptr[c] + i] == i;
                if (!same)
                    c = -1;
            }
            if (c >= 0)
            {
                P.alias_of[s] = c;
                P.ldf[s] = P.ldf[c];
                P.off[s] = P.off[c] + 6LL * P.ncb[c] * P.ldf[c] + 6LL * P.ncb[c];
                P.n_aliased++;
            }
            else
            {
                P.ldf[s] = 6LL * P.nb[s] + 1;
                P.off[s] = o;
                o += P.ldf[s] * 6LL * P.nb[s];
            }
        }
        P.front_doubles = o;
    }

    lap("5b storage");
    // ---- 6. assembly map of the Hsc blocks ---------------------------------------------
    const int B = rowptr[n];
    P.blk_front.resize(B), P.blk_row.resize(B), P.blk_col.resize(B), P.blk_trans.resize(B);
    for (int r = 0; r < n; r++)
        for (int k = rowptr[r]; k < rowptr[r + 1]; k++)
        {
            const int a = iperm[r], b = iperm[colind[k]];
            const int lo = std::min(a, b), hi = std::max(a, b);
            const int s = P.col_front[lo];
            int rowpos;
            if (hi < sfirst[s + 1])
                rowpos = hi - sfirst[s];
            else
            {
                const int* rb = P.rows.data() + P.rows_ptr[s];
                const int* re = P.rows.data() + P.rows_ptr[s + 1];
                const int* it = std::lower_bound(rb, re, hi);
                if (it == re || *it != hi)
                    throw std::runtime_error("cugo: symbolic: matrix entry outside front");
                rowpos = P.ncb[s] + (int)(it - rb);
            }
            P.blk_front[k] = s;
            P.blk_row[k] = rowpos;
            P.blk_col[k] = lo - sfirst[s];
            P.blk_trans[k] = (a < b) ? 1 : 0; // stored block is A(a,b); lower needs A(hi,lo)
        }

    lap("6 assembly map");
    // ---- 7. schedule: bottom subtrees -> one task each (stage 0), the rest by level ----
    std::vector<double> work(ns), sub(ns);
    for (int s = 0; s < ns; s++)
    {
        const double nc = 6.0 * P.ncb[s], nr = 6.0 * (P.nb[s] - P.ncb[s]);
        work[s] = nc * nc * nc / 3 + nc * nc * nr + nc * nr * nr + 2000.0;
        sub[s] = work[s];
    }
    for (int s = 0; s < ns; s++) // children have smaller index than parents (postorder)
        if (P.sparent[s] >= 0)
            sub[P.sparent[s]] += sub[s];
    double total = 0;
    for (int s = 0; s < ns; s++)
        total += work[s];
    const double wtask = total / std::max(1, opt.target_tasks);
    // a front is "lower" if its subtree fits in one task
    std::vector<char> lower(ns);
    for (int s = 0; s < ns; s++)
        lower[s] = sub[s] <= wtask;
    {
        int nroots = 0;
        for (int s = 0; s < ns; s++)
            if (lower[s] && (P.sparent[s] < 0 || !lower[P.sparent[s]]))
                nroots++;
        if (nroots < opt.min_subtree_tasks)
            std::fill(lower.begin(), lower.end(), 0);
    }
    std::vector<int> lvl(ns, 0);
    int max_lvl = 0;
    for (int s = 0; s < ns; s++)
    {
        if (lower[s])
            continue;
        int l = 1;
        for (int k = P.child_ptr[s]; k < P.child_ptr[s + 1]; k++)
        {
            const int c = P.child[k];
            l = std::max(l, lower[c] ? 1 : lvl[c] + 1);
        }
        lvl[s] = l;
        max_lvl = std::max(max_lvl, l);
    }
    // ---- 7a. ownership of elimination subtrees (world > 1): proportional mapping ------------------
    P.owner.assign(ns, opt.owned ? -1 : 0);
    P.xu_front.clear(), P.xu_stage.clear(), P.xu_owner.clear();
    P.xx_lo.clear(), P.xx_hi.clear(), P.xx_owner.clear();
    if (opt.owned)
    {
        // Candidates = subtrees that will go to one rank each; start from the roots of the forest.  While
        // the heaviest candidate is more than half of a rank's fair share, it is replaced by its children
        // (its root joins the replicated top): the candidates become many and small compared with a rank's
        // load.  Then largest first, each to the least loaded rank (ties: the lower rank).  Deterministic,
        // the same on every rank.
        double total_work = 0;
        for (int s = 0; s < ns; s++)
            total_work += work[s];
        const double fair = total_work / opt.world;
        std::vector<int> cand;
        for (int s = 0; s < ns; s++)
            if (P.sparent[s] < 0)
                cand.push_back(s);
        for (;;)
        {
            int big = -1;
            for (size_t i = 0; i < cand.size(); i++)
                if (P.child_ptr[cand[i] + 1] > P.child_ptr[cand[i]] && (big < 0 || sub[cand[i]] > sub[cand[big]]))
                    big = (int)i;
            if (big < 0 || sub[cand[big]] <= 0.5 * fair)
                break;
            const int v = cand[big];
            cand.erase(cand.begin() + big);
            P.owner[v] = -1; // (the default; spelled out: v is replicated)
            for (int k = P.child_ptr[v]; k < P.child_ptr[v + 1]; k++)
                cand.push_back(P.child[k]);
        }
        std::stable_sort(cand.begin(), cand.end(), [&](int a2, int b2) { return sub[a2] != sub[b2] ? sub[a2] > sub[b2] : a2 < b2; });
        std::vector<double> load(opt.world, 0.0);
        for (int v : cand)
        {
            const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            load[r] += sub[v];
            std::vector<int> st2{v};
            while (!st2.empty())
            {
                const int u = st2.back();
                st2.pop_back();
                P.owner[u] = r;
                for (int k = P.child_ptr[u]; k < P.child_ptr[u + 1]; k++)
                    st2.push_back(P.child[k]);
            }
        }
        for (int s = 0; s < ns; s++)
        {
            if (P.owner[s] < 0)
                P.top_flops += work[s];
            else if (P.owner[s] == opt.rank)
                P.rank_flops += work[s];
        }
    }
    else
        for (int s = 0; s < ns; s++)
            P.rank_flops += work[s];
    P.owned = opt.owned;
    P.xs_seg = P.xs_top = 0;
    P.xs_off.clear();
    if (opt.owned)
    { // layout of the ownership-keyed exchange (CholPlan::xs_off)
        const size_t B = P.blk_front.size();
        P.xs_off.assign(B + (size_t)n, 0);
        std::vector<int64_t> fill(opt.world + 1, 0); // [world] = the top
        auto slot = [&](int f) { return P.owner[f] < 0 ? opt.world : P.owner[f]; };
        for (size_t k = 0; k < B; k++)
        {
            const int q = slot(P.blk_front[k]);
            P.xs_off[k] = fill[q], fill[q] += 36;
        }
        for (int p = 0; p < n; p++)
        { // bsc row p lands in the rhs row of the front that holds column iperm[p]
            const int q = slot(P.col_front[P.iperm[p]]);
            P.xs_off[B + p] = fill[q], fill[q] += 6;
        }
        for (int r = 0; r < opt.world; r++)
            P.xs_seg = std::max(P.xs_seg, fill[r]);
        P.xs_seg = (P.xs_seg + 15) & ~int64_t(15); // (segments start on 128-byte boundaries)
        P.xs_top = fill[opt.world];
        for (size_t k = 0; k < B; k++)
            P.xs_off[k] += (int64_t)slot(P.blk_front[k]) * P.xs_seg;
        for (int p = 0; p < n; p++)
            P.xs_off[B + p] += (int64_t)slot(P.col_front[P.iperm[p]]) * P.xs_seg;
    }
    auto mine = [&](int s) { return P.owner[s] < 0 || P.owner[s] == opt.rank; };
    // stage 0 tasks: maximal lower subtrees (root = lower front whose parent is not lower)
    std::vector<int> task_root_of(ns, -1);
    std::vector<std::vector<int>> stage_tasks(max_lvl + 1);
    P.task_ptr.assign(1, 0);
    for (int s = ns - 1; s >= 0; s--)
    {
        if (!lower[s])
            continue;
        const int p = P.sparent[s];
        task_root_of[s] = (p >= 0 && lower[p]) ? task_root_of[p] : s;
    }
    {
        // collect fronts of each lower subtree in ascending (= post) order
        std::vector<int> roots;
        for (int s = 0; s < ns; s++)
            if (lower[s] && task_root_of[s] == s)
                roots.push_back(s);
        std::vector<int> root_slot(ns, -1);
        for (size_t i = 0; i < roots.size(); i++)
            root_slot[roots[i]] = (int)i;
        std::vector<std::vector<int>> lists(roots.size());
        for (int s = 0; s < ns; s++)
            if (lower[s])
                lists[root_slot[task_root_of[s]]].push_back(s);
        P.stage_task_ptr.assign(1, 0);
        for (auto& l : lists)
        {
            P.task_fronts.insert(P.task_fronts.end(), l.begin(), l.end());
            P.task_ptr.push_back((int)P.task_fronts.size());
        }
        if (!lists.empty())
            P.stage_task_ptr.push_back((int)P.task_ptr.size() - 1);
    }
    std::vector<int> stage_of(ns, 0);
    for (int l = 1; l <= max_lvl; l++)
    {
        bool any = false;
        for (int s = 0; s < ns; s++)
            if (!lower[s] && lvl[s] == l)
            {
                stage_of[s] = (int)P.stage_task_ptr.size() - 1; // (a level of the tree is a stage on EVERY rank,
                any = true;                                     // also one that holds none of this rank's fronts)
                if (!mine(s))
                    continue;
                P.task_fronts.push_back(s);
                P.task_ptr.push_back((int)P.task_fronts.size());
            }
        if (any)
            P.stage_task_ptr.push_back((int)P.task_ptr.size() - 1);
    }
    P.n_stages = (int)P.stage_task_ptr.size() - 1;
    if (opt.owned)
    {
        for (int s = 0; s < ns; s++)
            if (P.owner[s] >= 0 && P.sparent[s] >= 0 && P.owner[P.sparent[s]] < 0)
                P.xu_front.push_back(s), P.xu_stage.push_back(stage_of[s]), P.xu_owner.push_back(P.owner[s]);
        // maximal owned subtrees = maximal runs of columns solved by one rank
        for (int s = 0; s < ns; s++)
            if (P.owner[s] >= 0 && (P.sparent[s] < 0 || P.owner[P.sparent[s]] < 0))
            { // root of an owned subtree: its columns are those of the supernode range ending at s
                int first = s;
                std::vector<int> st2{s};
                while (!st2.empty())
                {
                    const int v = st2.back();
                    st2.pop_back();
                    first = std::min(first, v);
                    for (int k = P.child_ptr[v]; k < P.child_ptr[v + 1]; k++)
                        st2.push_back(P.child[k]);
                }
                P.xx_lo.push_back(P.col0[first]), P.xx_hi.push_back(P.col0[s] + P.ncb[s]), P.xx_owner.push_back(P.owner[s]);
            }
        // the ranges the solver will hand to the broadcast must lie inside the buffers (checked here, on the
        // host, where a mistake is an exception and not a device fault)
        for (size_t k = 0; k < P.xu_front.size(); k++)
        {
            const int f = P.xu_front[k];
            const int64_t ld = P.ldf[f], c0 = 6LL * P.ncb[f], c1 = 6LL * P.nb[f];
            const int64_t a = P.off[f] + c0 * ld + c0, b = a + (c1 - 1 - c0) * ld + c1 + 1 - c0;
            if (c1 > c0 && (a < 0 || b > P.front_doubles))
                throw std::runtime_error("cugo: symbolic: update block of front " + std::to_string(f) + " outside the front buffer");
        }
        for (size_t k = 0; k < P.xx_lo.size(); k++)
            if (P.xx_lo[k] < 0 || P.xx_hi[k] > n || P.xx_lo[k] >= P.xx_hi[k])
                throw std::runtime_error("cugo: symbolic: solution range of an owned subtree outside the matrix");
    }
    for (int s = 0; s < ns; s++)
    {
        const double nc = 6.0 * P.ncb[s], nr = 6.0 * (P.nb[s] - P.ncb[s]);
        P.backward_bytes += 8.0 * (nc * (nc + 1) / 2 + nc * nr + 3 * nc + nr);
    }
    P.has_subtree_stage = false;
    for (int s = 0; s < ns; s++)
        if (lower[s])
            P.has_subtree_stage = true;

    lap("7 schedule");
    // ---- 8. work items of the batched upper-stage kernels ------------------------------
    P.nc_max = 6;
    for (int s = 0; s < ns; s++)
        P.nc_max = std::max(P.nc_max, 6 * P.ncb[s]);
    P.ea_ptr.assign(1, 0), P.eab_ptr.assign(1, 0), P.syrk_ptr.assign(1, 0), P.bwg_ptr.assign(1, 0);
    P.lead_ptr.assign(1, 0), P.sb_ptr.assign(1, 0);
    P.bw_np.assign(ns, -1);
    // lead rows of a front: its leading boundary block rows inside the parent's pivot columns
    P.la_np.assign(ns, 0);
    for (int c = 0; c < ns; c++)
        if (P.sparent[c] >= 0)
        {
            const int nbr = P.nb[c] - P.ncb[c], ncbp = P.ncb[P.sparent[c]];
            int np = 0;
            while (np < nbr && P.rel[P.rel_ptr[c] + np] < ncbp)
                np++;
            P.la_np[c] = np;
        }
    std::vector<int32_t> lead, sb;
    P.l21off.assign(ns, -1);
    P.l21_doubles = 0;
    std::vector<int32_t> ea, eab, sy, bwg, tr;
    std::vector<int> sy_front_ntiles; // tiles per front of the level being listed
    P.trsm_ptr.assign(1, 0);
    P.stage_tile.assign(P.n_stages, 64);
    for (int st = 0; st < P.n_stages; st++)
    {
        const bool subtree = P.has_subtree_stage && st == 0;
        int TS = 64;
        bool two_phase = false;
        if (!subtree)
        { // tile edge of this level
            long n64 = 0;
            for (int t = P.stage_task_ptr[st]; t < P.stage_task_ptr[st + 1]; t++)
            {
                const int f = P.task_fronts[P.task_ptr[t]];
                const long nti = (6 * (P.nb[f] - P.ncb[f]) + 1 + 63) / 64;
                n64 += nti * (nti + 1) / 2;
            }
            if (n64 <= opt.tile32_max_tiles)
                TS = 32;
            P.stage_tile[st] = TS;
            // more tiles than CUs: the level is throughput-bound and the fused tile kernel recomputes every
            // L21 row tile once per tile of its row / column — solve them once instead (k_up_trsm, k_up_syrk)
            two_phase = opt.two_phase_min_tiles > 0 && n64 > opt.two_phase_min_tiles;
            if (two_phase)
                P.stage_tile[st] = 0;
        }
        if (!subtree)
            for (int t = P.stage_task_ptr[st]; t < P.stage_task_ptr[st + 1]; t++)
            {
                const int f = P.task_fronts[P.task_ptr[t]];
                const int nb = P.nb[f], ncb = P.ncb[f];
                {
                    const double nc = 6.0 * ncb, nr = 6.0 * (nb - ncb);
                    P.up_potrf_flops += nc * nc * nc / 3.0;
                    P.up_trsm_flops += nc * nc * (nr + 1);
                    P.up_syrk_flops += 2.0 * nc * (nr * (nr + 1) / 2 + nr);
                    for (int k = P.child_ptr[f]; k < P.child_ptr[f + 1] && P.alias_of[f] < 0; k++)
                    {
                        const double cr = 6.0 * (P.nb[P.child[k]] - P.ncb[P.child[k]]);
                        P.up_ea_bytes += 24.0 * (cr * (cr + 1) / 2 + cr); // read U, read+write parent
                    }
                }
                // extend-add work items: a workgroup has 16 waves and a unit is one 64-row chunk
                // of one block column, so an item takes as many block columns as fill 16 waves
                if (P.child_ptr[f + 1] > P.child_ptr[f] && P.alias_of[f] < 0)
                {
                    const int chunks = (6 * nb + 1 + 63) / 64;
                    const int cols = std::max(1, std::min(8, 16 / chunks));
                    for (int c0 = 0; c0 < ncb; c0 += cols)
                    {
                        ea.push_back(f), ea.push_back(c0), ea.push_back(std::min(ncb, c0 + cols));
                    }
                    for (int c0 = ncb; c0 < nb; c0 += cols)
                    {
                        eab.push_back(f), eab.push_back(c0), eab.push_back(std::min(nb, c0 + cols));
                    }
                }
                // fused trsm + syrk items (front, tile row ti, tile column tj) over the lower
                // triangle of 64x64 tiles of the update matrix (rows: boundary rows + rhs row);
                // tj = -1: a tile row without a diagonal tile (only the rhs row, or no boundary
                // at all) still needs its L21 rows solved and stored
                const int nrs = 6 * (nb - ncb), nbelow = nrs + 1;
                const int nti = (nbelow + TS - 1) / TS, ntj = (nrs + TS - 1) / TS;
                const size_t sy_front0 = sy.size();
                if (two_phase)
                { // syrk items (front, linear index over the lower triangle of tiles, tile columns), and
                  // the row tiles to solve first
                    int ntl = 0;
                    for (int tj = 0; tj < ntj; tj++)
                        ntl += nti - tj;
                    for (int t = 0; t < ntl; t++)
                    {
                        sy.push_back(f), sy.push_back(t), sy.push_back(ntj);
                    }
                    for (int r0 = 0; r0 < nbelow; r0 += 64)
                    {
                        tr.push_back(f), tr.push_back(r0), tr.push_back(std::min(64, nbelow - r0));
                    }
                }
                else
                {
                    for (int tj = 0; tj < ntj; tj++)
                        for (int ti = tj; ti < nti; ti++)
                        {
                            sy.push_back(f), sy.push_back(ti), sy.push_back(tj);
                        }
                    for (int ti = ntj; ti < nti; ti++)
                    {
                        sy.push_back(f), sy.push_back(ti), sy.push_back(-1);
                    }
                }
                sy_front_ntiles.push_back(std::max(1, (int)((sy.size() - sy_front0) / 3)));
                // look-ahead schedule: one lead workgroup per front that has lead rows, and the same
                // tiles without those that lie wholly inside the lead block
                const int q = 6 * P.la_np[f];
                if (q > 0)
                {
                    lead.push_back(f), lead.push_back(0), lead.push_back(0);
                }
                for (int tj = 0; tj < ntj; tj++)
                    for (int ti = tj; ti < nti; ti++)
                        if (TS * (ti + 1) > q)
                        {
                            sb.push_back(f), sb.push_back(ti), sb.push_back(tj);
                        }
                for (int ti = ntj; ti < nti; ti++)
                {
                    sb.push_back(f), sb.push_back(ti), sb.push_back(-1);
                }
                P.l21off[f] = P.l21_doubles;
                P.l21_doubles += (int64_t)6 * ncb * nbelow;
                // backward mat-vec items of this front's CHILDREN (upper-stage ones): rows beyond
                // the part owned by this front, 16 pivot columns per workgroup
                for (int k = P.child_ptr[f]; k < P.child_ptr[f + 1]; k++)
                {
                    const int c = P.child[k];
                    if (lower[c] || !mine(c))
                        continue;
                    const int nbr = P.nb[c] - P.ncb[c];
                    int npb = 0;
                    while (npb < nbr && P.rel[P.rel_ptr[c] + npb] < ncb)
                        npb++;
                    if (nbr - npb >= 8) // worth a workgroup only if a fair share is left
                    {
                        P.bw_np[c] = npb;
                        for (int j0 = 0; j0 < 6 * P.ncb[c]; j0 += 16)
                        {
                            bwg.push_back(c), bwg.push_back(j0), bwg.push_back(0);
                        }
                    }
                }
            }
        // XCD affinity: workgroups whose index is congruent mod 8 share an XCD and its L2 (observed
        // dispatch rule; a speed matter only).  All tiles of a front read the same W and the same L21
        // row panels.  So the fronts of the level are dealt to 8 groups (largest first, to the group with
        // the fewest tiles) and the tile items are interleaved so that the workgroup that runs item i is
        // in the group of i's front; a group that runs dry takes from the fullest.
        if (!subtree && opt.xcd_affinity && !sy_front_ntiles.empty())
        {
            const int nf = (int)sy_front_ntiles.size();
            int total = 0;
            for (int n : sy_front_ntiles)
                total += n;
            if (total > 8)
            {
                std::vector<int> byn(nf);
                for (int i = 0; i < nf; i++)
                    byn[i] = i;
                std::stable_sort(byn.begin(), byn.end(), [&](int a, int b) { return sy_front_ntiles[a] > sy_front_ntiles[b]; });
                std::vector<int> load(8, 0);
                std::vector<int> group_of(P.n_super, 0); // indexed by front id (only this level's are set)
                for (int i : byn)
                {
                    const int g = (int)(std::min_element(load.begin(), load.end()) - load.begin());
                    group_of[P.task_fronts[P.task_ptr[P.stage_task_ptr[st] + i]]] = g;
                    load[g] += sy_front_ntiles[i];
                }
                // items [first, end) of `list` (triples, front id first); the workgroup of item k has
                // index phase + k in its launch
                auto interleave = [&](std::vector<int32_t>& list, size_t first, int phase) {
                    const int n = (int)(list.size() / 3 - first);
                    if (n <= 1)
                        return;
                    std::vector<std::vector<int>> grp(8);
                    for (int k = 0; k < n; k++)
                        grp[group_of[list[3 * (first + k)]]].push_back(k);
                    std::vector<int32_t> out;
                    out.reserve((size_t)n * 3);
                    std::vector<size_t> pos(8, 0);
                    for (int k = 0; k < n; k++)
                    {
                        const int g = (phase + k) & 7;
                        int t;
                        if (pos[g] < grp[g].size())
                            t = grp[g][pos[g]++];
                        else
                        { // dry: the group with the most items left gives one (from its end)
                            int best = 0;
                            size_t left = 0;
                            for (int h = 0; h < 8; h++)
                                if (grp[h].size() - pos[h] > left)
                                    left = grp[h].size() - pos[h], best = h;
                            t = grp[best].back();
                            grp[best].pop_back();
                        }
                        out.insert(out.end(), list.begin() + 3 * (first + t), list.begin() + 3 * (first + t) + 3);
                    }
                    std::copy(out.begin(), out.end(), list.begin() + 3 * first);
                };
                // (the extend-add items of the potrf launch were tried the same way — they write what the
                // tiles read — and that took the gain away again: 14.67 vs 14.64 ms; not done)
                interleave(sy, (size_t)P.syrk_ptr.back(), 0);
                interleave(tr, (size_t)P.trsm_ptr.back(), 0);
            }
        }
        sy_front_ntiles.clear();
        P.ea_ptr.push_back((int)ea.size() / 3);
        P.eab_ptr.push_back((int)eab.size() / 3);
        P.syrk_ptr.push_back((int)sy.size() / 3);
        P.trsm_ptr.push_back((int)tr.size() / 3);
        P.bwg_ptr.push_back((int)bwg.size() / 3);
        P.lead_ptr.push_back((int)lead.size() / 3);
        P.sb_ptr.push_back((int)sb.size() / 3);
    }
    // one array: [ea | eab | trsyrk | backward mat-vec | lead | trsyrk without the lead tiles]; the ptr arrays
    // index items within their own section
    P.wl.clear();
    P.wl.insert(P.wl.end(), ea.begin(), ea.end());
    P.wl.insert(P.wl.end(), eab.begin(), eab.end());
    P.wl.insert(P.wl.end(), sy.begin(), sy.end());
    P.wl.insert(P.wl.end(), bwg.begin(), bwg.end());
    for (auto& v : P.eab_ptr)
        v += (int)ea.size() / 3;
    for (auto& v : P.syrk_ptr)
        v += (int)(ea.size() + eab.size()) / 3;
    for (auto& v : P.bwg_ptr)
        v += (int)(ea.size() + eab.size() + sy.size()) / 3;
    for (auto& v : P.trsm_ptr)
        v += (int)P.wl.size() / 3;
    P.wl.insert(P.wl.end(), tr.begin(), tr.end());
    // clearing the fronts before every assembly: only their lower triangles are ever written or read
    // for their value, so only those are cleared (items: front, first column, past-last column);
    // a front stored inside its child's update block has no storage of its own
    P.clr0 = (int)P.wl.size() / 3;
    P.nclr = 0;
    for (int f = 0; f < ns; f++)
        if (P.alias_of[f] < 0 && mine(f)) // (another rank's fronts are never read here, except through a broadcast)
            for (int c0 = 0; c0 < 6 * P.nb[f]; c0 += 16)
            {
                P.wl.push_back(f), P.wl.push_back(c0), P.wl.push_back(std::min(6 * P.nb[f], c0 + 16));
                P.nclr++;
            }
    // the assembly that makes that clear unnecessary: every lower-triangle entry of a stored front is written
    // once, with its Hsc entry (+ lambda), its right-hand side entry or zero
    P.asm0 = (int)P.wl.size() / 3;
    P.nasm = 0;
    P.asm_off.assign(ns, -1);
    P.asm_map.clear();
    {
        std::vector<int32_t> root(ns), shift(ns); // storage owner of a front, its origin inside (block rows)
        for (int f = 0; f < ns; f++)
        {
            int o = f;
            while (P.alias_of[o] >= 0)
                o = P.alias_of[o];
            const int64_t delta = P.off[f] - P.off[o], ld = P.ldf[o];
            if (P.ldf[f] != ld || delta % (ld + 1) != 0 || (delta / (ld + 1)) % 6 != 0 ||
                delta / (ld + 1) / 6 + P.nb[f] != P.nb[o])
                throw std::runtime_error("cugo: a front stored inside another one is not on its diagonal");
            root[f] = o, shift[f] = (int)(delta / (ld + 1) / 6);
        }
        for (int f = 0; f < ns; f++)
            if (P.alias_of[f] < 0 && mine(f))
            {
                const int64_t nb = P.nb[f];
                P.asm_off[f] = (int64_t)P.asm_map.size();
                P.asm_map.resize(P.asm_map.size() + nb + nb * (nb + 1) / 2, -1);
                for (int cb = 0; cb < nb; cb += 2)
                {
                    P.wl.push_back(f), P.wl.push_back(cb), P.wl.push_back((int)std::min<int64_t>(nb, cb + 2));
                    P.nasm++;
                }
            }
        for (int f = 0; f < ns; f++)
            if (P.asm_off[root[f]] >= 0)
                for (int c = 0; c < P.ncb[f]; c++)
                    P.asm_map[P.asm_off[root[f]] + shift[f] + c] = P.col0[f] + c;
        for (size_t k = 0; k < P.blk_front.size(); k++)
        {
            const int f = P.blk_front[k], o = root[f];
            if (P.asm_off[o] < 0)
                continue;
            const int64_t nb = P.nb[o], rb = P.blk_row[k] + shift[f], cb = P.blk_col[k] + shift[f];
            if (rb < cb || rb >= nb)
                throw std::runtime_error("cugo: an Hsc block outside the lower triangle of its front");
            int32_t& e = P.asm_map[P.asm_off[o] + nb + cb * nb - cb * (cb - 1) / 2 + (rb - cb)];
            if (e != -1)
                throw std::runtime_error("cugo: two Hsc blocks in one front position");
            e = 2 * (int32_t)k + (rb != cb && P.blk_trans[k] ? 1 : 0);
        }
    }
    for (auto& v : P.lead_ptr)
        v += (int)P.wl.size() / 3;
    P.wl.insert(P.wl.end(), lead.begin(), lead.end());
    for (auto& v : P.sb_ptr)
        v += (int)P.wl.size() / 3;
    P.wl.insert(P.wl.end(), sb.begin(), sb.end());
    lap("8 work items");
}

} // namespace cugo_host
