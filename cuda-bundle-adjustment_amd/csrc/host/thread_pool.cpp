// A small persistent host thread pool for the graph walks of initialize()/optimize().
// Those walks (edge objects, vertex objects, index sorts) are memory-latency bound and last
// 0.1-3 ms each on a 561k-edge graph; spawning 16 std::threads per walk costs about as much as
// the walk itself, so the workers are created once.  A worker spins for a short while after a walk
// (consecutive walks follow each other within microseconds) and parks on a condition variable then.
//
// Handing a walk to the workers takes no lock (round 4): with 63 workers picking the job up under one mutex
// the hand-over itself cost ~55 us per walk on the 64-thread pool of the MI355X box (tools/pool_latency.cpp: 88 us
// for a walk of 30 us chunks) — as much as the walk over the 133 k landmark objects of the kitti_00 shape, twice per
// BA call, and ~20 times per initialize() of a new graph.  Now a job lives in one of four slots; the caller fills the
// slot and publishes its generation g in ONE atomic word; a chunk is claimed by a compare-and-swap of its tag from
// 2g to 2g + 1, which can only succeed while the slot still holds job g — and job g cannot finish (so its slot
// cannot be re-used) while a claimed chunk is running.  The mutex is for parking and waking only.
#include "thread_pool.h"

#include <pthread.h>
#include <sched.h>

#include <cstdlib>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace cugo_host
{
namespace
{

constexpr unsigned kPoolCap = 64;    // workers (wanted_threads)
constexpr unsigned kMaxChunks = 256; // chunks of one job (more run in the caller, see pool_run)
constexpr unsigned kSlots = 4;

inline void relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
}

// one parallel walk
struct Job
{
    void (*fn)(void*, unsigned) = nullptr;
    void* ctx = nullptr;
    std::atomic<unsigned> chunks{0};
    std::atomic<uint64_t> tag[kMaxChunks]; // 2 g: chunk of job g, free; 2 g + 1: taken
    alignas(64) std::atomic<unsigned> done{0};

    // Thread `id` of `nthreads` first takes the chunks id, id + nthreads, ... — the same chunk
    // goes to the same thread in every walk, so what a thread touched in one walk (the vertex
    // objects of a landmark range, say) is still in its core's cache in the next — and then
    // whatever nobody has claimed yet (a worker that wakes up late delays nothing).
    // `n` is the number of chunks as THIS thread read it: if the slot has moved on to a later job the bound may be
    // that job's, which is harmless — every tag of a later job differs from 2 g and no claim succeeds.
    void work(uint64_t g, unsigned id, unsigned nthreads)
    {
        const unsigned n = std::min(chunks.load(std::memory_order_relaxed), kMaxChunks);
        for (unsigned c = id; c < n; c += nthreads)
            run(g, c);
        for (unsigned c = 0; c < n; c++)
            run(g, c);
    }
    void run(uint64_t g, unsigned c)
    {
        uint64_t expect = 2 * g;
        if (tag[c].load(std::memory_order_relaxed) != expect ||
            !tag[c].compare_exchange_strong(expect, 2 * g + 1, std::memory_order_acq_rel, std::memory_order_relaxed))
            return;
        // claimed: job g is alive until done reaches its chunk count, fn and ctx are its own (published before the tag)
        fn(ctx, c);
        done.fetch_add(1, std::memory_order_acq_rel);
    }
};

struct Pool
{
    std::mutex run_mutex; // one job at a time (callers on different host threads queue up)
    Job slot[kSlots];
    std::atomic<uint64_t> gen{0};     // generation of the job in slot[gen % kSlots]; 0: none yet
    std::atomic<uint64_t> prewake{0}; // calls of pool_prewake()
    std::atomic<int> parked{0};
    std::atomic<bool> stop{false};
    std::mutex m; // parking
    std::condition_variable cv_work;
    std::vector<std::thread> workers;

    void worker_main(unsigned id, unsigned nthreads)
    {
        uint64_t seen = 0, seen_pre = 0;
        int spin_us = 40;
        for (;;)
        {
            // consecutive walks follow each other within microseconds: spin briefly before parking (longer after a
            // pool_prewake(): the caller has announced a walk a few hundred microseconds ahead)
            const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
            uint64_t g = gen.load(std::memory_order_acquire);
            for (unsigned k = 0; g == seen; k++)
            {
                relax();
                if ((k & 15) == 15 && std::chrono::steady_clock::now() >= spin_until)
                    break;
                g = gen.load(std::memory_order_acquire);
            }
            spin_us = 40;
            if (g == seen)
            { // park (the check of gen / prewake and the count of parked workers pair with pool_run's store of gen
              // and its look at the count: sequentially consistent on both sides, one of the two sees the other)
                std::unique_lock<std::mutex> lk(m);
                parked.fetch_add(1, std::memory_order_seq_cst);
                cv_work.wait(lk, [&] {
                    return stop.load(std::memory_order_seq_cst) || gen.load(std::memory_order_seq_cst) != seen ||
                           prewake.load(std::memory_order_seq_cst) != seen_pre;
                });
                parked.fetch_sub(1, std::memory_order_seq_cst);
                if (stop.load())
                    return;
                seen_pre = prewake.load(std::memory_order_seq_cst);
                g = gen.load(std::memory_order_acquire);
                if (g == seen)
                { // only the announcement: stay up for the walk it announces
                    spin_us = 600;
                    continue;
                }
            }
            seen = g;
            slot[g % kSlots].work(g, id, nthreads);
        }
    }

    void wake_parked()
    {
        if (parked.load(std::memory_order_seq_cst) > 0)
        {
            { // (a worker between its check and its wait holds the mutex: it is waiting when we get it)
                std::lock_guard<std::mutex> lk(m);
            }
            cv_work.notify_all();
        }
    }
};

Pool* g_pool = nullptr;
std::once_flag g_once;
std::atomic<bool> g_forked{false};

// Width of the pool: the CPUs this process may run on (its affinity mask: a container's share of a 256-thread
// host, not the host), at most kPoolCap; CUGO_HOST_THREADS overrides.  The walks are memory-latency bound (edge and
// vertex objects scattered over the heap), so they scale with threads until the cores of the share are used up;
// measured on the MI355X box (EPYC 9575F): DESIGN.md section 6a.
unsigned wanted_threads()
{
    static const unsigned n = [] {
        unsigned cpus = std::thread::hardware_concurrency();
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
            cpus = std::min<unsigned>(cpus ? cpus : 1u, (unsigned)CPU_COUNT(&set));
        // one process per GPU on a node (torchrun sets LOCAL_WORLD_SIZE): the ranks share the CPUs of the mask
        if (const char* e = std::getenv("LOCAL_WORLD_SIZE"))
        {
            const int ranks = std::atoi(e);
            if (ranks > 1)
                cpus = std::max(1u, cpus / (unsigned)ranks);
        }
        unsigned want = std::min(kPoolCap, cpus);
        if (const char* e = std::getenv("CUGO_HOST_THREADS"))
            want = std::min(kPoolCap, (unsigned)std::max(1, std::atoi(e)));
        return std::max(1u, want);
    }();
    return n;
}

void create_pool()
{
    g_pool = new Pool;
    for (Job& j : g_pool->slot)
        for (auto& t : j.tag)
            t.store(~uint64_t(0), std::memory_order_relaxed);
    const unsigned n = wanted_threads();
    for (unsigned t = 1; t < n; t++)
        g_pool->workers.emplace_back([p = g_pool, t, n] { p->worker_main(t, n); });
    // a forked child has no worker threads: it runs every job in the calling thread
    pthread_atfork(nullptr, nullptr, [] { g_forked.store(true); });
}

} // namespace

unsigned pool_threads()
{
    return g_forked.load() ? 1u : wanted_threads();
}

// A parked worker needs 50 - 100 us to come back.  A caller that knows a walk is a few hundred microseconds away —
// optimize() before it downloads the estimates it is going to scatter — says so: the workers wake up now and spin
// until the walk arrives (at most 0.6 ms).
void pool_prewake()
{
    if (g_forked.load() || wanted_threads() == 1 || !g_pool)
        return;
    Pool& p = *g_pool;
    p.prewake.fetch_add(1, std::memory_order_seq_cst);
    p.wake_parked();
}

void pool_run(unsigned chunks, void (*fn)(void*, unsigned), void* ctx)
{
    if (chunks == 0)
        return;
    if (chunks == 1 || chunks > kMaxChunks || g_forked.load() || wanted_threads() == 1)
    {
        for (unsigned c = 0; c < chunks; c++)
            fn(ctx, c);
        return;
    }
    std::call_once(g_once, create_pool);
    Pool& p = *g_pool;
    std::lock_guard<std::mutex> run(p.run_mutex);
    const uint64_t g = p.gen.load(std::memory_order_relaxed) + 1;
    Job& job = p.slot[g % kSlots]; // (the job that used this slot, g - kSlots, finished before its pool_run returned)
    job.fn = fn, job.ctx = ctx;
    job.chunks.store(chunks, std::memory_order_relaxed);
    job.done.store(0, std::memory_order_relaxed);
    for (unsigned c = 0; c < chunks; c++)
        job.tag[c].store(2 * g, std::memory_order_release);
    p.gen.store(g, std::memory_order_seq_cst);
    p.wake_parked();
    job.work(g, 0, wanted_threads());
    for (unsigned k = 0; job.done.load(std::memory_order_acquire) < chunks; k++)
    {
        if (k < 4096)
            relax();
        else
            std::this_thread::yield();
    }
}

} // namespace cugo_host
