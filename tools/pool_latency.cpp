// Host only: what one pool_run costs by itself (empty chunks), with the workers spinning (back-to-back calls), parked
// (a pause of 2 ms before the call) and announced 100 us ahead (pool_prewake).
//   g++ -O2 -std=c++17 -pthread -I cuda-bundle-adjustment_amd/csrc/host tools/pool_latency.cpp cuda-bundle-adjustment_amd/csrc/host/thread_pool.cpp -o build_kp/pool_latency
#include "thread_pool.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
using Clock = std::chrono::steady_clock;
static double us(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
static void spin_us(double t)
{
    const auto t0 = Clock::now();
    while (us(t0, Clock::now()) < t)
    {
    }
}
int main()
{
    const unsigned n = cugo_host::pool_threads();
    std::printf("pool width %u\n", n);
    auto fn = [](void*, unsigned) {};
    auto work = [](void*, unsigned) { spin_us(30); }; // a chunk of a walk: 30 us
    for (int mode = 0; mode < 3; mode++)
        for (int kind = 0; kind < 2; kind++)
        {
            std::vector<double> t;
            for (int r = 0; r < 40; r++)
            {
                if (mode >= 1)
                    std::this_thread::sleep_for(std::chrono::milliseconds(2));
                if (mode == 2)
                {
                    cugo_host::pool_prewake();
                    spin_us(100);
                }
                const auto t0 = Clock::now();
                cugo_host::pool_run(n, kind ? +work : +fn, nullptr);
                t.push_back(us(t0, Clock::now()));
            }
            std::sort(t.begin(), t.end());
            std::printf("%-34s %-18s median %7.1f us  min %7.1f  max %7.1f\n",
                        mode == 0 ? "back to back (workers spinning)" : mode == 1 ? "after 2 ms (workers parked)" : "parked, announced 100 us ahead",
                        kind ? "30 us per chunk" : "empty chunks", t[t.size() / 2], t.front(), t.back());
        }
    return 0;
}
