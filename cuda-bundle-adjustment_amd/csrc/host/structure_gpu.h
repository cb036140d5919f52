// Device-side build of the Hsc block pattern + off-diagonal contribution lists (structure_gpu.cpp).
#pragma once
#include <cstdint>
#include <vector>

#include "../../../include/cugo_hip.h"
#include "hip_util.h"

namespace cugo_host
{

struct GpuStructureScratch
{
    DevBuf<uint64_t> npairs, pair_off, keys_a, keys_b, vals_a, vals_b;
    DevBuf<uint32_t> head, rank;
    DevBuf<int32_t> run_pa, run_pb, run_start, row_first;
    DevBuf<uint64_t> lrun_key;          // sharded build: pose-pair key of every LOCAL run
    DevBuf<int32_t> lrun_start, cov_ptr, cov_pose;
    DevBuf<char> temp;
    void release()
    {
        npairs.release(), pair_off.release(), keys_a.release(), keys_b.release(), vals_a.release(), vals_b.release();
        head.release(), rank.release(), run_pa.release(), run_pb.release(), run_start.release(), row_first.release();
        lrun_key.release(), lrun_start.release(), cov_ptr.release(), cov_pose.release();
        temp.release();
    }
};

struct GpuStructure
{
    // results on the device: upper block CSR of Hsc (diagonal block first in every row) and the
    // contribution lists of its off-diagonal blocks (cugo_hsc_struct of include/cugo_hip.h)
    DevBuf<int32_t> rowptr, colind, off_ptr, off_ei, off_ej;
    // the pattern on the host (symbolic analysis)
    std::vector<int32_t> h_rowptr, h_colind;
    int B = 0;
    size_t Moff = 0;   // products in the (local) lists
    size_t Mglobal = 0; // off-diagonal products of the whole graph (= Moff for a single process)
    GpuStructureScratch scratch; // transient; released by the caller when memory matters
    // two-phase build (build_pattern_gpu / build_lists_gpu): what the second phase needs from the first
    DevBuf<int32_t> run_pa, run_pb, row_first; // pose pair of every off-diagonal block of the pattern; first run of a row
    int bits = 1, n_runs = 0;
    GpuStructureScratch scratch2; // the second phase's own (the first may still be in flight on another stream)
};

// d_e_pose / d_flags: the flattened landmark-major edge slots, d_lm_ptr [Lall + 1].  Returns false
// (nothing built) when the problem is empty or the lists would not fit int32 indices.
// A shard passes the GLOBAL co-visibility as well (host arrays: free landmark -> its free poses over
// all shards, cov_ptr [L + 1]): the pattern then comes from those lists, the contribution lists from
// the local slots, and a block without local products gets an empty list.
bool build_structure_gpu(hipStream_t s, int E, int P, int Lall, const int32_t* d_e_pose, const uint8_t* d_flags,
                         const int32_t* d_lm_ptr, GpuStructure& out, int L = 0, const int32_t* h_cov_ptr = nullptr,
                         const int32_t* h_cov_pose = nullptr);

// The same build in two phases: the pattern from the co-visibility lists alone (host arrays, cov_ptr [L + 1]:
// free landmark -> its free poses over ALL shards), then the contribution lists from the local slots.
bool build_pattern_gpu(hipStream_t s, int P, int L, const int32_t* h_cov_ptr, const int32_t* h_cov_pose, GpuStructure& out);
bool build_lists_gpu(hipStream_t s, int E, int P, int Lall, const int32_t* d_e_pose, const uint8_t* d_flags,
                     const int32_t* d_lm_ptr, GpuStructure& out);

} // namespace cugo_host
