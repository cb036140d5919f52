// Device-side build of the Hsc block pattern and of the off-diagonal contribution lists.
// ref: HschurSparseBlockMatrix::constructFromVertices (src/sparse_block_matrix.cpp:63-156) and
// findHschureMulBlockIndicesKernel + thrust::sort (src/cuda/cuda_block_solver.cu:1347-1378,
// 1606-1634): the reference builds the product list on the device and sorts it by (i, j).
//
// Here, from the flattened landmark-major edge slots already on the device:
//   1. pairs per landmark (free-free active slots only), exclusive scan -> offsets, total M
//   2. one thread per landmark emits its pairs (a < b in slot order, hence pose(a) < pose(b)):
//      key = pose(a) << bits | pose(b), value = a << 32 | b
//   3. stable radix sort by key (rocPRIM): the pairs of an Hsc block become contiguous, in
//      ascending landmark order inside the block — the order the host build produces, so the
//      sums run in the same order and the results are bitwise the same
//   4. run heads + scan -> one run per off-diagonal block; rows by binary search; the diagonal
//      block of every free pose is inserted in front of its row
// The pattern (rowptr / colind, ~1 MB) goes back to the host for the symbolic analysis.
// A shard needs the GLOBAL pattern (the all-reduce payload has one layout on every rank), which its
// local edges do not give: it runs steps 1-4 twice — over the global co-visibility lists (keys only)
// for the pattern, over its local slots for the lists — and finds the list of every global block by
// binary search among its local runs (most blocks of a shard have few or no local products).
#include "structure_gpu.h"

#include <rocprim/rocprim.hpp>

namespace cugo_host
{

namespace
{
constexpr int TB = 256;

// flags == nullptr: every entry counts (the global co-visibility lists hold free-free edges only)
__device__ __forceinline__ bool slot_ff(const uint8_t* __restrict__ flags, int s)
{
    return !flags || (flags[s] & (CUGO_EDGE_FIXED_L | CUGO_EDGE_FIXED_P | CUGO_EDGE_INACTIVE)) == 0;
}

__global__ __launch_bounds__(TB) void k_count_pairs(int Lall, const int32_t* __restrict__ lm_ptr,
                                                    const uint8_t* __restrict__ flags,
                                                    uint64_t* __restrict__ npairs)
{
    const int l = blockIdx.x * TB + threadIdx.x;
    if (l >= Lall)
        return;
    uint64_t c = 0;
    for (int s = lm_ptr[l]; s < lm_ptr[l + 1]; s++)
        c += slot_ff(flags, s);
    npairs[l] = c * (c - (c > 0)) / 2;
}

__global__ __launch_bounds__(TB) void k_emit_pairs(int Lall, const int32_t* __restrict__ lm_ptr,
                                                   const uint8_t* __restrict__ flags,
                                                   const int32_t* __restrict__ pose, int bits,
                                                   const uint64_t* __restrict__ pair_off,
                                                   uint64_t* __restrict__ keys, uint64_t* __restrict__ vals)
{
    const int l = blockIdx.x * TB + threadIdx.x;
    if (l >= Lall)
        return;
    const int s0 = lm_ptr[l], s1 = lm_ptr[l + 1];
    uint64_t o = pair_off[l];
    for (int a = s0; a < s1; a++)
    {
        if (!slot_ff(flags, a))
            continue;
        const uint64_t pa = (uint64_t)pose[a] << bits;
        for (int b = a + 1; b < s1; b++)
            if (slot_ff(flags, b))
            {
                keys[o] = pa | (uint64_t)pose[b];
                if (vals)
                    vals[o] = ((uint64_t)a << 32) | (uint32_t)b;
                o++;
            }
    }
}

__global__ __launch_bounds__(TB) void k_heads(size_t M, const uint64_t* __restrict__ keys, uint32_t* __restrict__ head)
{
    const size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i < M)
        head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// one entry per run (off-diagonal block): its pose pair and the start of its list; and the split
// of the sorted values into the two edge-index arrays
__global__ __launch_bounds__(TB) void k_runs(size_t M, const uint64_t* __restrict__ keys,
                                             const uint64_t* __restrict__ vals, const uint32_t* __restrict__ head,
                                             const uint32_t* __restrict__ rank, int bits,
                                             int32_t* __restrict__ run_pa, int32_t* __restrict__ run_pb,
                                             int32_t* __restrict__ run_start, int32_t* __restrict__ off_ei,
                                             int32_t* __restrict__ off_ej, uint64_t* __restrict__ run_key)
{
    const size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i >= M)
        return;
    if (vals)
    {
        const uint64_t v = vals[i];
        off_ei[i] = (int32_t)(v >> 32);
        off_ej[i] = (int32_t)(v & 0xffffffffu);
    }
    if (head[i])
    {
        const uint32_t r = rank[i];
        const uint64_t k = keys[i];
        if (run_pa)
        {
            run_pa[r] = (int32_t)(k >> bits);
            run_pb[r] = (int32_t)(k & ((1ull << bits) - 1));
        }
        if (run_key)
            run_key[r] = k;
        if (run_start)
            run_start[r] = (int32_t)i;
    }
}

// sharded build: start of the local list of every global run = start of the first local run whose
// key is not smaller (an equal key: that run IS the block's list; a larger one: empty list)
__global__ __launch_bounds__(TB) void k_match_runs(int n_runs, const int32_t* __restrict__ run_pa,
                                                   const int32_t* __restrict__ run_pb, int bits, int n_lruns,
                                                   const uint64_t* __restrict__ lrun_key,
                                                   const int32_t* __restrict__ lrun_start, int32_t Mlocal,
                                                   int32_t* __restrict__ run_start)
{
    const int t = blockIdx.x * TB + threadIdx.x;
    if (t >= n_runs)
        return;
    const uint64_t key = ((uint64_t)run_pa[t] << bits) | (uint64_t)run_pb[t];
    int lo = 0, hi = n_lruns;
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (lrun_key[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    run_start[t] = lo < n_lruns ? lrun_start[lo] : Mlocal;
}

// rowptr[p] = (first run whose row is >= p) + p : every earlier row has one diagonal block in front
__global__ __launch_bounds__(TB) void k_rows(int P, int n_runs, const int32_t* __restrict__ run_pa,
                                             int32_t* __restrict__ row_first, int32_t* __restrict__ rowptr)
{
    const int p = blockIdx.x * TB + threadIdx.x;
    if (p > P)
        return;
    int lo = 0, hi = n_runs;
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (run_pa[mid] < p)
            lo = mid + 1;
        else
            hi = mid;
    }
    row_first[p] = lo;
    rowptr[p] = lo + p;
}

__global__ __launch_bounds__(TB) void k_blocks(int P, int n_runs, size_t M, const int32_t* __restrict__ run_pa,
                                               const int32_t* __restrict__ run_pb,
                                               const int32_t* __restrict__ run_start,
                                               const int32_t* __restrict__ row_first,
                                               const int32_t* __restrict__ rowptr, int32_t* __restrict__ colind,
                                               int32_t* __restrict__ off_ptr)
{
    // colind == nullptr: only the list starts (second phase); off_ptr == nullptr: only the pattern (first phase)
    const int t = blockIdx.x * TB + threadIdx.x;
    if (t < n_runs)
    { // off-diagonal block of run t
        const int id = t + run_pa[t] + 1;
        if (colind)
            colind[id] = run_pb[t];
        if (off_ptr)
            off_ptr[id] = run_start[t];
    }
    else if (t < n_runs + P)
    { // diagonal block of row p: first in its row, empty list
        const int p = t - n_runs;
        const int id = rowptr[p];
        const int r = row_first[p];
        if (colind)
            colind[id] = p;
        if (off_ptr)
            off_ptr[id] = r < n_runs ? run_start[r] : (int32_t)M;
    }
    else if (t == n_runs + P && off_ptr)
        off_ptr[n_runs + P] = (int32_t)M;
}
} // namespace

// steps 1-3 + run detection over one set of per-landmark lists; leaves the sorted keys (and values)
// in w.keys_b / w.vals_b, the run heads and ranks in w.head / w.rank.  Returns false on int32 overflow.
static bool sorted_pairs(hipStream_t s, int nlists, const int32_t* d_ptr, const uint8_t* d_flags, const int32_t* d_pose,
                         int bits, bool with_vals, GpuStructureScratch& w, size_t& M, int& n_runs)
{
    w.npairs.resize((size_t)nlists + 1), w.pair_off.resize((size_t)nlists + 1);
    CUGO_HIP(hipMemsetAsync(w.npairs.data() + nlists, 0, sizeof(uint64_t), s));
    hipLaunchKernelGGL(k_count_pairs, dim3((nlists + TB - 1) / TB), dim3(TB), 0, s, nlists, d_ptr, d_flags,
                       w.npairs.data());
    size_t tb = 0;
    CUGO_HIP(rocprim::exclusive_scan(nullptr, tb, w.npairs.data(), w.pair_off.data(), (uint64_t)0, (size_t)nlists + 1,
                                     rocprim::plus<uint64_t>(), s));
    w.temp.resize(tb + 16);
    CUGO_HIP(rocprim::exclusive_scan(w.temp.data(), tb, w.npairs.data(), w.pair_off.data(), (uint64_t)0,
                                     (size_t)nlists + 1, rocprim::plus<uint64_t>(), s));
    uint64_t M64 = 0;
    CUGO_HIP(hipMemcpyAsync(&M64, w.pair_off.data() + nlists, sizeof M64, hipMemcpyDeviceToHost, s));
    CUGO_HIP(hipStreamSynchronize(s));
    if (M64 >= (1ull << 31) - 1)
        return false; // the lists are indexed with int32 (as the host build's are)
    M = (size_t)M64;
    n_runs = 0;
    if (M == 0)
        return true;
    const int nM = (int)((M + TB - 1) / TB);
    w.keys_a.resize(M), w.keys_b.resize(M);
    if (with_vals)
        w.vals_a.resize(M), w.vals_b.resize(M);
    hipLaunchKernelGGL(k_emit_pairs, dim3((nlists + TB - 1) / TB), dim3(TB), 0, s, nlists, d_ptr, d_flags, d_pose, bits,
                       w.pair_off.data(), w.keys_a.data(), with_vals ? w.vals_a.data() : nullptr);
    tb = 0;
    if (with_vals)
    {
        CUGO_HIP(rocprim::radix_sort_pairs(nullptr, tb, w.keys_a.data(), w.keys_b.data(), w.vals_a.data(),
                                           w.vals_b.data(), M, 0u, (unsigned)(2 * bits), s));
        w.temp.resize(tb + 16);
        CUGO_HIP(rocprim::radix_sort_pairs(w.temp.data(), tb, w.keys_a.data(), w.keys_b.data(), w.vals_a.data(),
                                           w.vals_b.data(), M, 0u, (unsigned)(2 * bits), s));
    }
    else
    {
        CUGO_HIP(rocprim::radix_sort_keys(nullptr, tb, w.keys_a.data(), w.keys_b.data(), M, 0u, (unsigned)(2 * bits), s));
        w.temp.resize(tb + 16);
        CUGO_HIP(rocprim::radix_sort_keys(w.temp.data(), tb, w.keys_a.data(), w.keys_b.data(), M, 0u,
                                          (unsigned)(2 * bits), s));
    }
    w.head.resize(M + 1), w.rank.resize(M + 1);
    hipLaunchKernelGGL(k_heads, dim3(nM), dim3(TB), 0, s, M, w.keys_b.data(), w.head.data());
    CUGO_HIP(hipMemsetAsync(w.head.data() + M, 0, sizeof(uint32_t), s));
    tb = 0;
    CUGO_HIP(rocprim::exclusive_scan(nullptr, tb, w.head.data(), w.rank.data(), 0u, M + 1, rocprim::plus<uint32_t>(), s));
    w.temp.resize(tb + 16);
    CUGO_HIP(rocprim::exclusive_scan(w.temp.data(), tb, w.head.data(), w.rank.data(), 0u, M + 1,
                                     rocprim::plus<uint32_t>(), s));
    uint32_t nr = 0;
    CUGO_HIP(hipMemcpyAsync(&nr, w.rank.data() + M, sizeof nr, hipMemcpyDeviceToHost, s));
    CUGO_HIP(hipStreamSynchronize(s));
    n_runs = (int)nr;
    return true;
}

bool build_structure_gpu(hipStream_t s, int E, int P, int Lall, const int32_t* d_e_pose, const uint8_t* d_flags,
                         const int32_t* d_lm_ptr, GpuStructure& out, int L, const int32_t* h_cov_ptr,
                         const int32_t* h_cov_pose)
{
    out.B = 0, out.Moff = 0, out.Mglobal = 0;
    if (P <= 0 || Lall <= 0 || E < 0)
        return false;
    const bool sharded = h_cov_ptr != nullptr;
    if (!sharded && E == 0)
        return false;
    int bits = 1;
    while ((1ll << bits) < (long long)P + 1)
        bits++;
    GpuStructureScratch& w = out.scratch;
    // ---- the local slots: contribution lists (and, single process, the pattern)
    size_t M = 0;
    int n_lruns = 0;
    if (E > 0 && !sorted_pairs(s, Lall, d_lm_ptr, d_flags, d_e_pose, bits, true, w, M, n_lruns))
        return false;
    const int nM = (int)((M + TB - 1) / TB);
    out.off_ei.resize(M + 16), out.off_ej.resize(M + 16);
    int n_runs = n_lruns;
    size_t Mg = M;
    if (!sharded)
    {
        w.run_pa.resize((size_t)n_runs + 1), w.run_pb.resize((size_t)n_runs + 1), w.run_start.resize((size_t)n_runs + 1);
        if (M > 0)
            hipLaunchKernelGGL(k_runs, dim3(nM), dim3(TB), 0, s, M, w.keys_b.data(), w.vals_b.data(), w.head.data(),
                               w.rank.data(), bits, w.run_pa.data(), w.run_pb.data(), w.run_start.data(),
                               out.off_ei.data(), out.off_ej.data(), (uint64_t*)nullptr);
    }
    else
    {
        w.lrun_key.resize((size_t)n_lruns + 1), w.lrun_start.resize((size_t)n_lruns + 1);
        if (M > 0)
            hipLaunchKernelGGL(k_runs, dim3(nM), dim3(TB), 0, s, M, w.keys_b.data(), w.vals_b.data(), w.head.data(),
                               w.rank.data(), bits, (int32_t*)nullptr, (int32_t*)nullptr, w.lrun_start.data(),
                               out.off_ei.data(), out.off_ej.data(), w.lrun_key.data());
        // ---- the global co-visibility lists: the pattern
        const size_t ncov = (size_t)h_cov_ptr[L];
        w.cov_ptr.resize((size_t)L + 1), w.cov_pose.resize(ncov + 1);
        CUGO_HIP(hipMemcpyAsync(w.cov_ptr.data(), h_cov_ptr, sizeof(int32_t) * ((size_t)L + 1), hipMemcpyHostToDevice, s));
        if (ncov > 0)
            CUGO_HIP(hipMemcpyAsync(w.cov_pose.data(), h_cov_pose, sizeof(int32_t) * ncov, hipMemcpyHostToDevice, s));
        if (!sorted_pairs(s, L, w.cov_ptr.data(), nullptr, w.cov_pose.data(), bits, false, w, Mg, n_runs))
            return false;
        w.run_pa.resize((size_t)n_runs + 1), w.run_pb.resize((size_t)n_runs + 1), w.run_start.resize((size_t)n_runs + 1);
        if (Mg > 0)
        {
            hipLaunchKernelGGL(k_runs, dim3((unsigned)((Mg + TB - 1) / TB)), dim3(TB), 0, s, Mg, w.keys_b.data(),
                               (const uint64_t*)nullptr, w.head.data(), w.rank.data(), bits, w.run_pa.data(),
                               w.run_pb.data(), (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                               (uint64_t*)nullptr);
            hipLaunchKernelGGL(k_match_runs, dim3((n_runs + TB - 1) / TB), dim3(TB), 0, s, n_runs, w.run_pa.data(),
                               w.run_pb.data(), bits, n_lruns, w.lrun_key.data(), w.lrun_start.data(), (int32_t)M,
                               w.run_start.data());
        }
    }
    const int B = n_runs + P;
    out.rowptr.resize((size_t)P + 1), out.colind.resize((size_t)B + 1), out.off_ptr.resize((size_t)B + 1);
    w.row_first.resize((size_t)P + 1);
    hipLaunchKernelGGL(k_rows, dim3((P + 1 + TB - 1) / TB), dim3(TB), 0, s, P, n_runs, w.run_pa.data(),
                       w.row_first.data(), out.rowptr.data());
    hipLaunchKernelGGL(k_blocks, dim3((B + 1 + TB - 1) / TB), dim3(TB), 0, s, P, n_runs, M, w.run_pa.data(),
                       w.run_pb.data(), w.run_start.data(), w.row_first.data(), out.rowptr.data(), out.colind.data(),
                       out.off_ptr.data());
    out.h_rowptr.resize((size_t)P + 1), out.h_colind.resize((size_t)B);
    CUGO_HIP(hipMemcpyAsync(out.h_rowptr.data(), out.rowptr.data(), sizeof(int32_t) * ((size_t)P + 1),
                            hipMemcpyDeviceToHost, s));
    CUGO_HIP(hipMemcpyAsync(out.h_colind.data(), out.colind.data(), sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    CUGO_HIP(hipStreamSynchronize(s));
    CUGO_HIP(hipGetLastError());
    out.B = B, out.Moff = M, out.Mglobal = Mg;
    return true;
}

// ---- the same build in two phases (the engine overlaps the first with the rest of initialize()) ----
// Phase 1, from the co-visibility lists alone (host arrays: free landmark -> its free poses, cov_ptr [L + 1]):
// the Hsc pattern.  It is known as soon as the edges have been grouped by landmark — before the slot layout,
// the slot arrays and their upload exist — and it is all the ordering + symbolic analysis needs.
bool build_pattern_gpu(hipStream_t s, int P, int L, const int32_t* h_cov_ptr, const int32_t* h_cov_pose, GpuStructure& out)
{
    out.B = 0, out.Moff = 0, out.Mglobal = 0, out.n_runs = 0;
    if (P <= 0 || L < 0)
        return false;
    int bits = 1;
    while ((1ll << bits) < (long long)P + 1)
        bits++;
    out.bits = bits;
    GpuStructureScratch& w = out.scratch;
    const size_t ncov = (size_t)h_cov_ptr[L];
    w.cov_ptr.resize((size_t)L + 1), w.cov_pose.resize(ncov + 1);
    CUGO_HIP(hipMemcpyAsync(w.cov_ptr.data(), h_cov_ptr, sizeof(int32_t) * ((size_t)L + 1), hipMemcpyHostToDevice, s));
    if (ncov > 0)
        CUGO_HIP(hipMemcpyAsync(w.cov_pose.data(), h_cov_pose, sizeof(int32_t) * ncov, hipMemcpyHostToDevice, s));
    size_t Mg = 0;
    int n_runs = 0;
    if (L > 0 && !sorted_pairs(s, L, w.cov_ptr.data(), nullptr, w.cov_pose.data(), bits, false, w, Mg, n_runs))
        return false;
    out.run_pa.resize((size_t)n_runs + 1), out.run_pb.resize((size_t)n_runs + 1);
    if (Mg > 0)
        hipLaunchKernelGGL(k_runs, dim3((unsigned)((Mg + TB - 1) / TB)), dim3(TB), 0, s, Mg, w.keys_b.data(),
                           (const uint64_t*)nullptr, w.head.data(), w.rank.data(), bits, out.run_pa.data(),
                           out.run_pb.data(), (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                           (uint64_t*)nullptr);
    const int B = n_runs + P;
    out.rowptr.resize((size_t)P + 1), out.colind.resize((size_t)B + 1), out.row_first.resize((size_t)P + 1);
    hipLaunchKernelGGL(k_rows, dim3((P + 1 + TB - 1) / TB), dim3(TB), 0, s, P, n_runs, out.run_pa.data(),
                       out.row_first.data(), out.rowptr.data());
    hipLaunchKernelGGL(k_blocks, dim3((B + 1 + TB - 1) / TB), dim3(TB), 0, s, P, n_runs, (size_t)0, out.run_pa.data(),
                       out.run_pb.data(), (const int32_t*)nullptr, out.row_first.data(), out.rowptr.data(),
                       out.colind.data(), (int32_t*)nullptr);
    out.h_rowptr.resize((size_t)P + 1), out.h_colind.resize((size_t)B);
    CUGO_HIP(hipMemcpyAsync(out.h_rowptr.data(), out.rowptr.data(), sizeof(int32_t) * ((size_t)P + 1),
                            hipMemcpyDeviceToHost, s));
    CUGO_HIP(hipMemcpyAsync(out.h_colind.data(), out.colind.data(), sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    CUGO_HIP(hipStreamSynchronize(s));
    CUGO_HIP(hipGetLastError());
    out.B = B, out.Mglobal = Mg, out.n_runs = n_runs;
    return true;
}

// Phase 2, from the flattened landmark-major edge slots on the device: the contribution lists of the pattern's
// blocks (stable sort: ascending landmark order inside a block, the host build's order), matched to the
// pattern's runs by binary search — a block without local products (a shard) gets an empty list.
bool build_lists_gpu(hipStream_t s, int E, int P, int Lall, const int32_t* d_e_pose, const uint8_t* d_flags,
                     const int32_t* d_lm_ptr, GpuStructure& out)
{
    GpuStructureScratch& w = out.scratch2;
    const int bits = out.bits, n_runs = out.n_runs, B = out.B;
    size_t M = 0;
    int n_lruns = 0;
    if (E > 0 && !sorted_pairs(s, Lall, d_lm_ptr, d_flags, d_e_pose, bits, true, w, M, n_lruns))
        return false;
    const int nM = (int)((M + TB - 1) / TB);
    out.off_ei.resize(M + 16), out.off_ej.resize(M + 16);
    w.lrun_key.resize((size_t)n_lruns + 1), w.lrun_start.resize((size_t)n_lruns + 1);
    w.run_start.resize((size_t)n_runs + 1);
    if (M > 0)
        hipLaunchKernelGGL(k_runs, dim3(nM), dim3(TB), 0, s, M, w.keys_b.data(), w.vals_b.data(), w.head.data(),
                           w.rank.data(), bits, (int32_t*)nullptr, (int32_t*)nullptr, w.lrun_start.data(),
                           out.off_ei.data(), out.off_ej.data(), w.lrun_key.data());
    if (n_runs > 0)
        hipLaunchKernelGGL(k_match_runs, dim3((n_runs + TB - 1) / TB), dim3(TB), 0, s, n_runs, out.run_pa.data(),
                           out.run_pb.data(), bits, n_lruns, w.lrun_key.data(), w.lrun_start.data(), (int32_t)M,
                           w.run_start.data());
    out.off_ptr.resize((size_t)B + 1);
    hipLaunchKernelGGL(k_blocks, dim3((B + 1 + TB - 1) / TB), dim3(TB), 0, s, P, n_runs, M, out.run_pa.data(),
                       out.run_pb.data(), w.run_start.data(), out.row_first.data(), out.rowptr.data(),
                       (int32_t*)nullptr, out.off_ptr.data());
    CUGO_HIP(hipStreamSynchronize(s));
    CUGO_HIP(hipGetLastError());
    out.Moff = M;
    return true;
}

} // namespace cugo_host
