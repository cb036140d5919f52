// Landmark-major plan of the Schur complement's H-side (Hsc = Hpp - sum T Hpl^T).
//
// The reference forms one 6x6 product per thread and adds it to its Hsc block with 36 atomics
// (ref: findHschureMulBlockIndicesKernel / computeHschureKernel, src/cuda/cuda_block_solver.cu:
// 1327-1378).  The gather kernels of round 1 walk the products destination-major and re-fetch
// the operands from L2 / Infinity Cache (4.3x the algorithmic bytes).  This plan walks them
// SOURCE-major: a workgroup owns 256 consecutive edge slots (= whole landmarks in the engine's
// padded layout), holds their Hpl and T blocks in LDS, and forms all products of its landmarks
// there.  Products of one group that share a destination block are summed in the group (fixed
// order) into a "partial slot"; a second pass adds the slots of every Hsc block in group order.
// Every Hpl block is read from HBM once and T is never written.
#pragma once
#include <cstdint>
#include <vector>

#include "../../../include/cugo_hip.h"

namespace cugo_host
{

constexpr int kSchurGroup = 256; // edge slots per group (= the workgroup size of the edge kernels)
constexpr int kSchurLongSlot = 12; // a slot with more products is worked on by a whole wave (k_schur_fused)

struct SchurPlanHost
{
    bool usable = false;          // false: a landmark's edges straddle two groups -> gather kernels
    int n_groups = 0, n_slots = 0, n_rhs = 0;
    std::vector<int32_t> grp_ptr;  // [n_groups+1] partial slots of each group (longest product list first)
    std::vector<int32_t> grp_nwave; // [n_groups] leading slots with more than kSchurLongSlot products
    std::vector<int32_t> slot_blk; // [n_slots] Hsc block of the slot
    std::vector<int32_t> slot_rhs; // [n_slots] rhs partial index (diagonal blocks) or -1
    std::vector<int32_t> slot_ptr; // [n_slots+1] product range of the slot
    std::vector<uint16_t> prod;    // [M] a | b << 8: local slots of the T and the Hpl operand
    std::vector<int32_t> red_ptr;  // [B+1] partial slots of every Hsc block ...
    std::vector<int32_t> red_slot; // [n_slots] ... in group order
    std::vector<int32_t> blk_pose; // [B] free pose of a diagonal block, -1 for an off-diagonal one
    double bytes() const;          // device bytes of the partial buffers
};

// e_pose / e_lm / flags: the flattened landmark-major edge slots (E of them); rowptr / colind: the
// Hsc pattern (upper block CSR, diagonal first).  Threaded (thread_pool.h).
void build_schur_plan(int E, int P, const int32_t* e_pose, const int32_t* e_lm, const uint8_t* flags,
                      const int32_t* rowptr, const int32_t* colind, SchurPlanHost& out);

} // namespace cugo_host

#include "hip_util.h"

namespace cugo_host
{
// the plan on the device + the partial buffers; fill() points the plan fields of a cugo_hsc_struct at it
struct SchurPlanDevice
{
    DevBuf<int32_t> grp_ptr, grp_nwave, slot_rhs, slot_ptr, red_ptr, red_slot, blk_pose;
    DevBuf<uint16_t> prod;
    DevBuf<double> part_H, part_b;
    int n_groups = 0, n_slots = 0, n_rhs = 0;
    void upload(const SchurPlanHost& h, hipStream_t s); // synchronises the stream before returning
    void fill(cugo_hsc_struct& hs) const;
    static void clear(cugo_hsc_struct& hs);
};

} // namespace cugo_host
