// RCCL binding (see rccl_comm.h).  The all-reduce of [Hsc | bsc] and of the LM scalars runs on
// the solver's own stream: nothing synchronises with the host between the Schur kernels, the
// collective and the factorisation.
#include "rccl_comm.h"

#include <dlfcn.h>

#include <mutex>
#include <stdexcept>
#include <string>

#include <rccl/rccl.h>

namespace cugo_host
{

namespace
{
struct Api
{
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclReduceScatter) reduce_scatter = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

Api& api()
{
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        // an already loaded librccl (e.g. the one inside a PyTorch wheel) is found by its soname
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
                break;
        if (!a.handle)
            return;
        a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(a.handle, "ncclGetUniqueId"));
        a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(a.handle, "ncclCommInitRank"));
        a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(a.handle, "ncclCommDestroy"));
        a.all_reduce = reinterpret_cast<decltype(a.all_reduce)>(dlsym(a.handle, "ncclAllReduce"));
        a.broadcast = reinterpret_cast<decltype(a.broadcast)>(dlsym(a.handle, "ncclBroadcast"));
        a.reduce_scatter = reinterpret_cast<decltype(a.reduce_scatter)>(dlsym(a.handle, "ncclReduceScatter"));
        a.group_start = reinterpret_cast<decltype(a.group_start)>(dlsym(a.handle, "ncclGroupStart"));
        a.group_end = reinterpret_cast<decltype(a.group_end)>(dlsym(a.handle, "ncclGroupEnd"));
        a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(a.handle, "ncclGetErrorString"));
    });
    if (!a.handle || !a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_reduce)
        throw std::runtime_error("cugo: librccl.so could not be loaded (multi-GPU runs need RCCL)");
    return a;
}

void check(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess)
    {
        const Api& a = api();
        throw std::runtime_error(std::string("cugo: RCCL ") + what + " failed: " +
                                 (a.error_string ? a.error_string(r) : "error"));
    }
}
} // namespace

void RcclComm::unique_id(void* id128)
{
    static_assert(sizeof(ncclUniqueId) == kRcclUniqueIdBytes, "unique id size");
    check(api().get_unique_id(static_cast<ncclUniqueId*>(id128)), "ncclGetUniqueId");
}

RcclComm::RcclComm(const void* id128, int rank, int world) : rank_(rank), world_(world)
{
    ncclUniqueId id;
    __builtin_memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    check(api().comm_init_rank(&c, world, id, rank), "ncclCommInitRank");
    comm_ = c;
}

RcclComm::~RcclComm()
{
    if (comm_)
        (void)api().comm_destroy(static_cast<ncclComm_t>(comm_));
}

void RcclComm::all_reduce(double* d_buf, size_t n, int op, hipStream_t s)
{
    check(api().all_reduce(d_buf, d_buf, n, ncclDouble, op == 0 ? ncclSum : ncclMax,
                           static_cast<ncclComm_t>(comm_), s),
          "ncclAllReduce");
}

void RcclComm::broadcast(double* d_buf, size_t n, int root, hipStream_t s)
{
    if (!api().broadcast)
        throw std::runtime_error("cugo: librccl.so has no ncclBroadcast");
    check(api().broadcast(d_buf, d_buf, n, ncclDouble, root, static_cast<ncclComm_t>(comm_), s), "ncclBroadcast");
}

void RcclComm::reduce_scatter(double* d_buf, size_t n_seg, hipStream_t s)
{
    if (!api().reduce_scatter)
        throw std::runtime_error("cugo: librccl.so has no ncclReduceScatter");
    // in place: the receive buffer is this rank's segment of the send buffer (rccl.h: recvbuff == sendbuff + rank * recvcount)
    check(api().reduce_scatter(d_buf, d_buf + (size_t)rank_ * n_seg, n_seg, ncclDouble, ncclSum,
                               static_cast<ncclComm_t>(comm_), s),
          "ncclReduceScatter");
}

// the broadcasts between group(true) and group(false) are issued as ONE fused operation (ncclGroupStart / End):
// the update blocks that cross the ownership boundary after a level have different roots and sizes
void RcclComm::group(bool start)
{
    if (!api().group_start || !api().group_end)
        return; // (each broadcast then is an operation of its own)
    if (start)
        check(api().group_start(), "ncclGroupStart");
    else
        check(api().group_end(), "ncclGroupEnd");
}

} // namespace cugo_host
