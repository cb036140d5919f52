// GPU box: does a kernel always see what the kernel before it (same stream) has written — also when the writers
// ran on other XCDs and shared cache lines with each other?  (the rare run-to-run deviation
// reads like a consumer seeing the value from BEFORE its producer's write.)  The access pattern of the sparse
// factorisation in miniature: short kernels back to back, few workgroups, every workgroup writes a short run of
// 8-byte words next to its neighbours' (workgroups with consecutive indices run on different XCDs, so the lines are
// shared between XCDs), the next kernel's workgroups read what OTHER workgroups wrote.  Every word carries the
// iteration it was written in; a reader that finds an older iteration counts it.
//   hipcc --offload-arch=gfx950 -O2 tools/coherence_selftest.hip -o /tmp/coherence_selftest && /tmp/coherence_selftest [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x)                                                       \
    do                                                              \
    {                                                               \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess)                                       \
        {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(2);                                                \
        }                                                           \
    } while (0)

// workgroup b writes words [b * run, (b + 1) * run)
__global__ void k_write(uint64_t* __restrict__ a, int run, uint64_t it)
{
    for (int i = threadIdx.x; i < run; i += blockDim.x)
    {
        const uint64_t idx = (uint64_t)blockIdx.x * run + i;
        a[idx] = (it << 24) | idx;
    }
}

// workgroup b reads the runs of workgroups b + 1, b + 3 and b + nblk / 2 (+ 1), i.e. words it did not write itself
__global__ void k_read(const uint64_t* __restrict__ a, int run, int nwriters, uint64_t it, uint64_t* __restrict__ err)
{
    const int shifts[3] = {1, 3, nwriters / 2 + 1};
    for (int s = 0; s < 3; s++)
    {
        const int w = (int)((blockIdx.x + (unsigned)shifts[s]) % (unsigned)nwriters);
        for (int i = threadIdx.x; i < run; i += blockDim.x)
        {
            const uint64_t idx = (uint64_t)w * run + i;
            const uint64_t v = a[idx];
            if (v != ((it << 24) | idx))
            {
                const unsigned long long k = atomicAdd((unsigned long long*)err, 1ULL);
                if (k < 8)
                { // what was read, where, by whom
                    err[1 + 4 * k] = v, err[2 + 4 * k] = (it << 24) | idx;
                    err[3 + 4 * k] = blockIdx.x, err[4 + 4 * k] = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xF;
                }
            }
        }
    }
}

// Second pattern (a front's pivot block is READ by its potrf workgroup while other workgroups of the same launch add
// the children's entries to the rows below it — same cache lines; the next launch reads those rows): line b =
// 16 words.  k_mixed: workgroup b reads word 0 of its line at once (the line now sits in its XCD's L2), and — late —
// writes word 8 of workgroup b+1's line.  k_check (next launch): workgroup b reads word 8 of ITS line.
__global__ void k_mixed(uint64_t* __restrict__ a, uint64_t it, uint64_t* __restrict__ sink)
{
    const unsigned b = blockIdx.x, n = gridDim.x;
    if (threadIdx.x == 0)
        sink[b] = a[16 * b];
    if (threadIdx.x == 64)
    {
        for (int k = 0; k < 12; k++)
            __builtin_amdgcn_s_sleep(100);
        a[16 * ((b + 1) % n) + 8] = it;
    }
}
__global__ void k_check(const uint64_t* __restrict__ a, uint64_t it, uint64_t* __restrict__ err)
{
    if (threadIdx.x == 0)
    {
        const uint64_t v = a[16 * blockIdx.x + 8];
        if (v != it)
        {
            const unsigned long long k = atomicAdd((unsigned long long*)err, 1ULL);
            if (k < 8)
            {
                err[1 + 4 * k] = v << 24, err[2 + 4 * k] = it << 24;
                err[3 + 4 * k] = blockIdx.x, err[4 + 4 * k] = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xF;
            }
        }
    }
}

// Third pattern (an update block is read-modify-written by the extend-add workgroups of one launch, by the tile
// workgroups of the next, ... each time from another XCD): launch number k lets workgroup b add 1 to every word of
// the run of workgroup b + k.  After K launches every word must hold K: a read-modify-write that started from a
// stale value loses increments for good.
__global__ void k_rmw(uint64_t* __restrict__ a, int run, unsigned shift)
{
    const unsigned w = (blockIdx.x + shift) % gridDim.x;
    for (int i = threadIdx.x; i < run; i += blockDim.x)
        a[(uint64_t)w * run + i] += 1;
}
__global__ void k_rmw_check(const uint64_t* __restrict__ a, long n, uint64_t due, uint64_t* __restrict__ err)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (a[i] != due)
        {
            const unsigned long long k = atomicAdd((unsigned long long*)err, 1ULL);
            if (k < 8)
                err[1 + 4 * k] = a[i], err[2 + 4 * k] = due, err[3 + 4 * k] = (uint64_t)i, err[4 + 4 * k] = 0;
        }
}

int main(int argc, char** argv)
{
    const long iters = argc > 1 ? atol(argv[1]) : 30000;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint64_t *d_a, *d_err;
    CK(hipMalloc(&d_a, 4096 * 1024 * 8));
    CK(hipMalloc(&d_err, 64 * 8));
    long total_bad = 0;
    const int nblks[] = {2, 8, 24, 64, 256, 1024};
    const int runs[] = {6, 8, 37, 97, 512};
    const int nreaders_of[] = {0, 1}; // 0: as many readers as writers, 1: ONE reading workgroup (the root fronts' case)
    for (int one_reader : nreaders_of)
        for (int nblk : nblks)
            for (int run : runs)
            {
                CK(hipMemsetAsync(d_err, 0, 64 * 8, st));
                CK(hipMemsetAsync(d_a, 0xFF, (size_t)nblk * run * 8, st));
                const int threads = run >= 256 ? 256 : 64;
                for (long it = 1; it <= iters; it++)
                {
                    k_write<<<nblk, threads, 0, st>>>(d_a, run, (uint64_t)it);
                    k_read<<<one_reader ? 1 : nblk, threads, 0, st>>>(d_a, run, nblk, (uint64_t)it, d_err);
                }
                CK(hipGetLastError());
                uint64_t h[64];
                CK(hipMemcpyAsync(h, d_err, sizeof h, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                if (h[0])
                {
                    total_bad += (long)h[0];
                    printf("%4d writers x %3d words, %s: %llu stale words in %ld write/read pairs\n", nblk, run,
                           one_reader ? "one reader" : "as many readers", (unsigned long long)h[0], iters);
                    for (int k = 0; k < 8 && k < (int)h[0]; k++)
                        printf("    read iteration %llu word %llu where iteration %llu was due (reader workgroup %llu on xcc %llu)\n",
                               (unsigned long long)(h[1 + 4 * k] >> 24), (unsigned long long)(h[1 + 4 * k] & 0xFFFFFF),
                               (unsigned long long)(h[2 + 4 * k] >> 24), (unsigned long long)h[3 + 4 * k],
                               (unsigned long long)h[4 + 4 * k]);
                }
            }
    uint64_t* d_sink;
    CK(hipMalloc(&d_sink, 4096 * 8));
    for (int nblk : {2, 8, 24, 64, 256, 1024, 4096})
    {
        CK(hipMemsetAsync(d_err, 0, 64 * 8, st));
        CK(hipMemsetAsync(d_a, 0, (size_t)nblk * 16 * 8, st));
        for (long it = 1; it <= iters; it++)
        {
            k_mixed<<<nblk, 128, 0, st>>>(d_a, (uint64_t)it, d_sink);
            k_check<<<nblk, 64, 0, st>>>(d_a, (uint64_t)it, d_err);
        }
        CK(hipGetLastError());
        uint64_t h[64];
        CK(hipMemcpyAsync(h, d_err, sizeof h, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        if (h[0])
        {
            total_bad += (long)h[0];
            printf("read-beside-write pattern, %4d lines: %llu stale words in %ld launch pairs\n", nblk, (unsigned long long)h[0], iters);
            for (int k = 0; k < 8 && k < (int)h[0]; k++)
                printf("    read iteration %llu where iteration %llu was due (workgroup %llu on xcc %llu)\n",
                       (unsigned long long)(h[1 + 4 * k] >> 24), (unsigned long long)(h[2 + 4 * k] >> 24),
                       (unsigned long long)h[3 + 4 * k], (unsigned long long)h[4 + 4 * k]);
        }
    }
    for (int nblk : {2, 8, 24, 64, 256})
        for (int run : {6, 37, 97, 512})
        {
            CK(hipMemsetAsync(d_err, 0, 64 * 8, st));
            CK(hipMemsetAsync(d_a, 0, (size_t)nblk * run * 8, st));
            for (long it = 1; it <= 2 * iters; it++)
                k_rmw<<<nblk, run >= 256 ? 256 : 64, 0, st>>>(d_a, run, (unsigned)(it * 5 % nblk));
            k_rmw_check<<<64, 256, 0, st>>>(d_a, (long)nblk * run, (uint64_t)(2 * iters), d_err);
            CK(hipGetLastError());
            uint64_t h[64];
            CK(hipMemcpyAsync(h, d_err, sizeof h, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            if (h[0])
            {
                total_bad += (long)h[0];
                printf("read-modify-write chain, %4d workgroups x %3d words: %llu words short after %ld launches\n", nblk, run,
                       (unsigned long long)h[0], 2 * iters);
                for (int k = 0; k < 8 && k < (int)h[0]; k++)
                    printf("    word %llu holds %llu, due %llu\n", (unsigned long long)h[3 + 4 * k], (unsigned long long)h[1 + 4 * k],
                           (unsigned long long)h[2 + 4 * k]);
            }
        }
    printf("coherence self-test: %ld stale words in total (%ld write/read kernel pairs per shape, 60 + 7 + 20 shapes)\n", total_bad, iters);
    return 0;
}
