// Process-wide cache of device (and pinned host) allocations.
// ORB-SLAM2-style callers create a new optimiser for every bundle adjustment, as they do with
// g2o; hipMalloc / hipFree cost 50-100 us each and an optimiser owns ~40 buffers, so on a
// local-BA sized graph allocation and release took longer (2.5 ms to destroy, ~1 ms to set up)
// than the ten LM iterations.  Freed blocks go to per-size free lists instead and are handed out
// again; sizes are rounded up to 1/8-octave classes (<= 12.5 % slack).  CUGO_POOL_MAX_MB caps
// the cached bytes (default 16384; 0 disables the cache).
#include "hip_util.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace cugo_host
{
namespace
{

struct Block
{
    size_t bytes;
    int device;
    bool pinned;
};

struct Cache
{
    std::mutex m;
    // (device, pinned, class bytes) -> free blocks
    std::map<std::tuple<int, bool, size_t>, std::vector<void*>> free_lists;
    std::unordered_map<void*, Block> live; // blocks handed out by the cache
    size_t cached_bytes = 0;
    size_t max_cached = 0;
    std::map<int, std::vector<hipStream_t>> idle_streams; // per device
    Cache()
    {
        const char* e = std::getenv("CUGO_POOL_MAX_MB");
        max_cached = (e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)16384) << 20;
    }
};

Cache& cache()
{
    static Cache* c = new Cache; // never destroyed: the HIP runtime may be gone before static destructors run
    return *c;
}

size_t size_class(size_t bytes)
{
    size_t c = 512;
    if (bytes <= c)
        return c;
    // largest power of two <= bytes, then steps of 1/8 of it
    size_t p = 1;
    while ((p << 1) <= bytes)
        p <<= 1;
    const size_t step = std::max<size_t>(p >> 3, 512);
    return (bytes + step - 1) / step * step;
}

} // namespace

void* cache_alloc(size_t bytes, bool pinned, size_t* got_bytes)
{
    Cache& c = cache();
    const size_t cls = size_class(bytes);
    int dev = 0;
    CUGO_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(c.m);
        auto it = c.free_lists.find({dev, pinned, cls});
        if (it != c.free_lists.end() && !it->second.empty())
        {
            void* p = it->second.back();
            it->second.pop_back();
            c.cached_bytes -= cls;
            c.live[p] = {cls, dev, pinned};
            *got_bytes = cls;
            return p;
        }
    }
    void* p = nullptr;
    if (pinned)
        // coherent (fine-grained) on purpose, whatever HIP_HOST_COHERENT says: the host polls words of these blocks
        // that a kernel writes with a system-scope release (Engine::optimize, the trial's sequence number)
        CUGO_HIP(hipHostMalloc(&p, cls, hipHostMallocCoherent));
    else
        CUGO_HIP(hipMalloc(&p, cls));
    {
        std::lock_guard<std::mutex> lk(c.m);
        c.live[p] = {cls, dev, pinned};
    }
    *got_bytes = cls;
    return p;
}

// CUGO_POISON_ALLOC=1: the two guard zones of a device buffer must still hold their pattern
void guard_check(const void* raw, size_t payload_bytes, bool floating, const char* what)
{
    (void)hipDeviceSynchronize();
    unsigned char g[2][kGuardBytes];
    if (hipMemcpy(g[0], raw, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(g[1], static_cast<const char*>(raw) + kGuardBytes + payload_bytes, kGuardBytes,
                  hipMemcpyDeviceToHost) != hipSuccess)
        return;
    const unsigned char want = floating ? (unsigned char)poison_byte() : 0x00;
    for (int side = 0; side < 2; side++)
        for (size_t i = 0; i < kGuardBytes; i++)
            if (g[side][i] != want)
            {
                std::fprintf(stderr,
                             "cugo: guard zone %s a device buffer of %zu bytes was overwritten at byte %zu (%s)\n",
                             side ? "behind" : "before", payload_bytes, i, what);
                std::abort();
            }
}

void cache_free(void* p)
{
    if (!p)
        return;
    Cache& c = cache();
    Block b{0, 0, false};
    bool keep = false;
    {
        std::lock_guard<std::mutex> lk(c.m);
        auto it = c.live.find(p);
        if (it == c.live.end())
            return; // not ours (cannot happen)
        b = it->second;
        c.live.erase(it);
        keep = c.cached_bytes + b.bytes <= c.max_cached;
    }
    if (keep)
    {
        // hipFree waits for the device; a cached block may be handed out again at once, so wait
        // here as well (cheap when the device is idle, which is the normal case: optimisers
        // synchronise their stream before they release anything)
        (void)hipDeviceSynchronize();
        std::lock_guard<std::mutex> lk(c.m);
        c.free_lists[{b.device, b.pinned, b.bytes}].push_back(p);
        c.cached_bytes += b.bytes;
        return;
    }
    if (b.pinned)
        (void)hipHostFree(p);
    else
        (void)hipFree(p);
}

// Creating / destroying a stream costs 1-2 ms (a hardware queue is set up): the optimisers of
// successive BA calls share a few non-blocking streams instead.  A released stream is idle (its
// owner has synchronised it).
hipStream_t cache_stream_acquire()
{
    Cache& c = cache();
    int dev = 0;
    CUGO_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(c.m);
        auto& v = c.idle_streams[dev];
        if (!v.empty())
        {
            hipStream_t s = v.back();
            v.pop_back();
            return s;
        }
    }
    hipStream_t s = nullptr;
    CUGO_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}

void cache_stream_release(hipStream_t s)
{
    if (!s)
        return;
    Cache& c = cache();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return;
    std::lock_guard<std::mutex> lk(c.m);
    auto& v = c.idle_streams[dev];
    if (v.size() < 16 && c.max_cached > 0)
        v.push_back(s);
    else
        (void)hipStreamDestroy(s);
}

} // namespace cugo_host
