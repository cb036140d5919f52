// Persistent host thread pool for the graph walks of initialize()/optimize() (thread_pool.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

namespace cugo_host
{

unsigned pool_threads(); // chunks worth asking for: the CPUs of the share of this process (<= 64; CUGO_HOST_THREADS)
// fn(ctx, c) for c in [0, chunks), spread over the pool; returns when all chunks are done
void pool_run(unsigned chunks, void (*fn)(void*, unsigned), void* ctx);
// announces a pool_run a few hundred microseconds ahead: parked workers wake up and wait for it spinning
void pool_prewake();

// f(begin, end, chunk) over [0, n) in contiguous chunks; a single chunk in the caller below
// `serial_below` items.  Returns the number of chunks used.
template <typename F>
inline unsigned parallel_chunks(size_t n, size_t serial_below, F&& f)
{
    const unsigned nt = n < serial_below ? 1u : pool_threads();
    if (nt <= 1)
    {
        f((size_t)0, n, 0u);
        return 1;
    }
    struct Ctx
    {
        F* f;
        size_t n;
        unsigned nt;
    } ctx{&f, n, nt};
    pool_run(
        nt,
        [](void* p, unsigned c) {
            Ctx& x = *static_cast<Ctx*>(p);
            (*x.f)(x.n * c / x.nt, x.n * (c + 1) / x.nt, c);
        },
        &ctx);
    return nt;
}

} // namespace cugo_host
