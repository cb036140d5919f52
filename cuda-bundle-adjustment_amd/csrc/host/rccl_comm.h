// RCCL communicator of a landmark-sharded run (one process per GPU, xGMI inside the node).
// librccl is loaded on first use (dlopen): a single-GPU user never pays for it and the
// library has no link-time dependency on it.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace cugo_host
{

constexpr int kRcclUniqueIdBytes = 128; // NCCL_UNIQUE_ID_BYTES

class RcclComm
{
public:
    // rank 0 of a job: a fresh id (ncclGetUniqueId) that the caller hands to every rank
    static void unique_id(void* id128);
    // collective over all ranks of the job: ncclCommInitRank on the CURRENT HIP device
    RcclComm(const void* id128, int rank, int world);
    ~RcclComm();
    RcclComm(const RcclComm&) = delete;
    RcclComm& operator=(const RcclComm&) = delete;
    // in-place all-reduce of n doubles on `s` (op 0 = sum, 1 = max); asynchronous, no host sync
    void all_reduce(double* d_buf, size_t n, int op, hipStream_t s);
    // in-place broadcast of n doubles from rank `root` on `s`; asynchronous
    void broadcast(double* d_buf, size_t n, int root, hipStream_t s);
    // in-place sum reduce-scatter of world() segments of n_seg doubles starting at d_buf: afterwards segment rank()
    // of THIS rank holds the sum over ranks of that segment (the other segments: unspecified); asynchronous
    void reduce_scatter(double* d_buf, size_t n_seg, hipStream_t s);
    // broadcasts issued between group(true) and group(false) form one fused operation
    void group(bool start);
    int rank() const { return rank_; }
    int world() const { return world_; }

private:
    void* comm_ = nullptr;
    int rank_ = 0, world_ = 1;
};

} // namespace cugo_host
