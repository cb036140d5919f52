// see schur_plan.h
#include "schur_plan.h"

#include <algorithm>
#include <cstring>

#include "thread_pool.h"

namespace cugo_host
{

double SchurPlanHost::bytes() const { return 8.0 * (36.0 * n_slots + 6.0 * n_rhs); }

void SchurPlanDevice::upload(const SchurPlanHost& h, hipStream_t s)
{
    n_groups = h.n_groups, n_slots = h.n_slots, n_rhs = h.n_rhs;
    grp_ptr.upload(h.grp_ptr, s), grp_nwave.upload(h.grp_nwave, s), slot_rhs.upload(h.slot_rhs, s), slot_ptr.upload(h.slot_ptr, s);
    red_ptr.upload(h.red_ptr, s), red_slot.upload(h.red_slot, s), blk_pose.upload(h.blk_pose, s);
    prod.upload(h.prod, s);
    part_H.resize(36 * (size_t)h.n_slots + 16), part_b.resize(6 * (size_t)h.n_rhs + 16);
    CUGO_HIP(hipStreamSynchronize(s));
}

void SchurPlanDevice::fill(cugo_hsc_struct& hs) const
{
    hs.n_groups = n_groups, hs.n_slots = n_slots, hs.n_rhs = n_rhs;
    hs.d_grp_ptr = grp_ptr.data(), hs.d_grp_nwave = grp_nwave.data(), hs.d_slot_rhs = slot_rhs.data(), hs.d_slot_ptr = slot_ptr.data();
    hs.d_prod = prod.data(), hs.d_red_ptr = red_ptr.data(), hs.d_red_slot = red_slot.data();
    hs.d_blk_pose = blk_pose.data(), hs.d_part_H = part_H.data(), hs.d_part_b = part_b.data();
}

void SchurPlanDevice::clear(cugo_hsc_struct& hs)
{
    hs.n_groups = hs.n_slots = hs.n_rhs = 0;
    hs.d_grp_ptr = hs.d_grp_nwave = hs.d_slot_rhs = hs.d_slot_ptr = hs.d_red_ptr = hs.d_red_slot = hs.d_blk_pose = nullptr;
    hs.d_prod = nullptr;
    hs.d_part_H = hs.d_part_b = nullptr;
}

namespace
{
struct GroupOut
{
    std::vector<int32_t> blk, rhs, cnt; // per slot: Hsc block, is-diagonal flag, number of products
    std::vector<uint16_t> prod;
    bool ok = true;
};
} // namespace

void build_schur_plan(int E, int P, const int32_t* e_pose, const int32_t* e_lm, const uint8_t* flags,
                      const int32_t* rowptr, const int32_t* colind, SchurPlanHost& out)
{
    out = SchurPlanHost{};
    const int G = (E + kSchurGroup - 1) / kSchurGroup;
    const int B = P > 0 ? rowptr[P] : 0;
    out.n_groups = G;
    out.blk_pose.assign(B, -1);
    for (int p = 0; p < P; p++)
        out.blk_pose[rowptr[p]] = p; // the diagonal block is the first of its row
    auto free_free = [&](int s) {
        return (flags[s] & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE)) == 0;
    };
    // a landmark whose active edges lie in two groups cannot be handled group-locally
    for (int g = 1; g < G && G > 0; g++)
    {
        const int s = g * kSchurGroup;
        if (e_lm[s] != e_lm[s - 1])
            continue;
        bool before = false, after = false;
        for (int i = s - 1; i >= 0 && e_lm[i] == e_lm[s]; i--)
            before = before || free_free(i);
        for (int i = s; i < E && e_lm[i] == e_lm[s]; i++)
            after = after || free_free(i);
        if (before && after)
            return; // usable stays false
    }
    std::vector<GroupOut> go(G);
    parallel_chunks((size_t)G, 64, [&](size_t ga, size_t gb, unsigned) {
        struct Item
        {
            int32_t blk;
            uint16_t ab;
        };
        std::vector<Item> items;
        std::vector<int32_t> run, runs;
        std::vector<int> ord;
        for (size_t g = ga; g < gb; g++)
        {
            GroupOut& o = go[g];
            const int s0 = (int)g * kSchurGroup, s1 = std::min(E, s0 + kSchurGroup);
            items.clear();
            int i = s0;
            while (i < s1)
            {
                int j = i;
                run.clear();
                while (j < s1 && e_lm[j] == e_lm[i])
                {
                    if (free_free(j))
                        run.push_back(j);
                    j++;
                }
                // slots of a landmark are sorted by pose index: (a, b) with a <= b is an upper block
                for (size_t x = 0; x < run.size(); x++)
                {
                    const int pa = e_pose[run[x]];
                    const int32_t* c0 = colind + rowptr[pa];
                    const int32_t* c1 = colind + rowptr[pa + 1];
                    for (size_t y = x; y < run.size(); y++)
                    {
                        const int pb = e_pose[run[y]];
                        int32_t k;
                        if (pb == pa)
                            k = rowptr[pa];
                        else
                        { // columns behind the diagonal block are ascending
                            const int32_t* it = std::lower_bound(c0 + 1, c1, pb);
                            if (it == c1 || *it != pb)
                            {
                                o.ok = false; // not in the pattern: cannot happen with a consistent structure
                                continue;
                            }
                            k = (int32_t)(it - colind);
                        }
                        items.push_back({k, (uint16_t)((run[x] - s0) | ((run[y] - s0) << 8))});
                    }
                }
                i = j;
            }
            // by destination block; inside a block the products keep ascending landmark order
            std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.blk < b.blk; });
            // runs of equal block = partial slots; the slots of a group are emitted longest first:
            // the kernel deals them to its lane groups in this order, the long lists start first
            runs.clear();
            for (size_t q = 0; q < items.size(); q++)
                if (q == 0 || items[q].blk != items[q - 1].blk)
                    runs.push_back((int32_t)q);
            runs.push_back((int32_t)items.size());
            const int nr = (int)runs.size() - 1;
            ord.resize(nr);
            for (int q = 0; q < nr; q++)
                ord[q] = q;
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) {
                return runs[x + 1] - runs[x] > runs[y + 1] - runs[y];
            });
            o.prod.reserve(items.size());
            for (int q = 0; q < nr; q++)
            {
                const int a = runs[ord[q]], b = runs[ord[q] + 1];
                o.blk.push_back(items[a].blk);
                o.rhs.push_back(out.blk_pose[items[a].blk] >= 0 ? 1 : 0);
                o.cnt.push_back(b - a);
                for (int z = a; z < b; z++)
                    o.prod.push_back(items[z].ab);
            }
        }
    });
    size_t nslots = 0, nprod = 0;
    out.grp_ptr.assign(G + 1, 0);
    out.grp_nwave.assign(G, 0);
    for (int g = 0; g < G; g++)
    {
        if (!go[g].ok)
            return;
        for (int32_t c : go[g].cnt)
            out.grp_nwave[g] += c > kSchurLongSlot;
        nslots += go[g].blk.size();
        nprod += go[g].prod.size();
        out.grp_ptr[g + 1] = (int32_t)nslots;
    }
    out.n_slots = (int)nslots;
    out.slot_blk.resize(nslots), out.slot_rhs.resize(nslots), out.slot_ptr.assign(nslots + 1, 0);
    out.prod.resize(nprod);
    int nrhs = 0;
    size_t sp = 0, pp = 0;
    for (int g = 0; g < G; g++)
    {
        const GroupOut& o = go[g];
        for (size_t q = 0; q < o.blk.size(); q++, sp++)
        {
            out.slot_blk[sp] = o.blk[q];
            out.slot_rhs[sp] = o.rhs[q] ? nrhs++ : -1;
            out.slot_ptr[sp + 1] = out.slot_ptr[sp] + o.cnt[q];
        }
        if (!o.prod.empty())
            std::memcpy(out.prod.data() + pp, o.prod.data(), o.prod.size() * sizeof(uint16_t));
        pp += o.prod.size();
    }
    out.n_rhs = nrhs;
    // slots of every Hsc block, ascending group (counting sort keeps the order)
    out.red_ptr.assign(B + 1, 0);
    for (size_t q = 0; q < nslots; q++)
        out.red_ptr[out.slot_blk[q] + 1]++;
    for (int k = 0; k < B; k++)
        out.red_ptr[k + 1] += out.red_ptr[k];
    out.red_slot.resize(nslots);
    {
        std::vector<int32_t> fill(out.red_ptr.begin(), out.red_ptr.end() - 1);
        for (size_t q = 0; q < nslots; q++)
            out.red_slot[fill[out.slot_blk[q]]++] = (int32_t)q;
    }
    out.usable = true;
}

} // namespace cugo_host
