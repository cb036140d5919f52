// A small persistent host thread pool for the graph walks of initialize()/optimize().
// Those walks (edge objects, vertex objects, index sorts) are memory-latency bound and last
// 0.5-3 ms each on a 561k-edge graph; spawning 16 std::threads per walk costs about as much as
// the walk itself, so the workers are created once and parked on a condition variable.
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

// one parallel walk; workers hold a reference, so a worker that wakes up late finds its own
// (fully claimed) job and never touches the state of the next one
struct Job
{
    void (*fn)(void*, unsigned) = nullptr;
    void* ctx = nullptr;
    unsigned chunks = 0;
    std::unique_ptr<std::atomic<uint8_t>[]> taken;
    std::atomic<unsigned> done{0};

    // Thread `id` of `nthreads` first takes the chunks id, id + nthreads, ... — the same chunk
    // goes to the same thread in every walk, so what a thread touched in one walk (the vertex
    // objects of a landmark range, say) is still in its core's cache in the next — and then
    // whatever nobody has claimed yet (a worker that wakes up late delays nothing).
    void work(unsigned id, unsigned nthreads)
    {
        for (unsigned c = id; c < chunks; c += nthreads)
            run(c);
        for (unsigned c = 0; c < chunks; c++)
            run(c);
    }
    void run(unsigned c)
    {
        if (taken[c].load(std::memory_order_relaxed) || taken[c].exchange(1, std::memory_order_acq_rel))
            return;
        fn(ctx, c);
        done.fetch_add(1, std::memory_order_acq_rel);
    }
};

struct Pool
{
    std::mutex run_mutex; // one job at a time (callers on different host threads queue up)
    std::mutex m;
    std::condition_variable cv_work;
    std::vector<std::thread> workers;
    uint64_t generation = 0;      // guarded by m
    std::shared_ptr<Job> current; // guarded by m
    std::atomic<uint64_t> gen_hint{0}; // lock-free copy of `generation` for the short spin
    bool stop = false;

    void worker_main(unsigned id, unsigned nthreads)
    {
        uint64_t seen = 0;
        for (;;)
        {
            // consecutive walks follow each other within microseconds: spin briefly before parking
            const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(40);
            while (gen_hint.load(std::memory_order_acquire) == seen &&
                   std::chrono::steady_clock::now() < spin_until)
            {
            }
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop || generation != seen; });
                if (stop)
                    return;
                seen = generation;
                job = current;
            }
            job->work(id, nthreads);
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
constexpr unsigned kPoolCap = 64;
unsigned wanted_threads()
{
    static const unsigned n = [] {
        unsigned cpus = std::thread::hardware_concurrency();
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
            cpus = std::min<unsigned>(cpus ? cpus : 1u, (unsigned)CPU_COUNT(&set));
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

void pool_run(unsigned chunks, void (*fn)(void*, unsigned), void* ctx)
{
    if (chunks == 0)
        return;
    if (chunks == 1 || g_forked.load() || wanted_threads() == 1)
    {
        for (unsigned c = 0; c < chunks; c++)
            fn(ctx, c);
        return;
    }
    std::call_once(g_once, create_pool);
    Pool& p = *g_pool;
    std::lock_guard<std::mutex> run(p.run_mutex);
    auto job = std::make_shared<Job>();
    job->fn = fn, job->ctx = ctx, job->chunks = chunks;
    job->taken.reset(new std::atomic<uint8_t>[chunks]);
    for (unsigned c = 0; c < chunks; c++)
        job->taken[c].store(0, std::memory_order_relaxed);
    {
        std::lock_guard<std::mutex> lk(p.m);
        p.current = job;
        p.generation++;
        p.gen_hint.store(p.generation, std::memory_order_release);
    }
    p.cv_work.notify_all();
    job->work(0, wanted_threads());
    while (job->done.load(std::memory_order_acquire) < chunks)
        std::this_thread::yield();
}

} // namespace cugo_host
