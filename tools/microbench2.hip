// Second calibration pass for the Cholesky kernels: UNROLLED dependent chains (the loops of
// tools/microbench.hip carry ~32 cycles of loop overhead per iteration, which is what its
// "dependent fp64 fma: 40 ticks" actually measured), issue cost of independent fp64 ops of one
// wave, cross-lane broadcast primitives, LDS broadcast reads, barrier cost inside a loop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench2 tools/microbench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ void k_fma_dep(int n, long long* cyc, double* out)
{
    double x = threadIdx.x * 1e-3, y = 1.0000001;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < U; u++)
            x = fma(x, y, 1e-9);
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = x;
}
template <int U>
__global__ void k_mul_dep(int n, long long* cyc, double* out)
{
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0000001;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < U; u++)
            x = x * y;
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = x;
}
// NCH independent chains, unrolled 16 deep
template <int NCH>
__global__ void k_fma_ind(int n, long long* cyc, double* out)
{
    double x[NCH];
    for (int i = 0; i < NCH; i++)
        x[i] = threadIdx.x * 1e-3 + i;
    double y = 1.0000001;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int i = 0; i < NCH; i++)
                x[i] = fma(x[i], y, 1e-9);
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    double s = 0;
    for (int i = 0; i < NCH; i++)
        s += x[i];
    out[threadIdx.x] = s;
}
// rsq + coupled Newton (the sqrt / 1/sqrt pair of the panel factorisation), dependent
__global__ void k_rsq_newton(int n, long long* cyc, double* out)
{
    double d = 2.0 + threadIdx.x * 1e-3;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            const double y = __builtin_amdgcn_rsq(d);
            double g = d * y, h = 0.5 * y;
            const double r0 = fma(-h, g, 0.5);
            const double h2 = fma(y, r0, y);
            g = fma(g, r0, g), h = fma(h, r0, h);
            const double dg = fma(-g, g, d), rh = fma(-h, g, 0.5);
            const double sq = fma(dg, h, g);
            const double inv = fma(h2, rh, h2);
            d = fma(sq, inv, 1.5); // next "pivot" depends on both
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = d;
}
// bare v_rsq_f64 chain
__global__ void k_rsq_dep(int n, long long* cyc, double* out)
{
    double d = 2.0 + threadIdx.x * 1e-3;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
            d = __builtin_amdgcn_rsq(d) + 1.0;
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = d;
}
// readlane broadcast of a double + dependent fma (cross-lane chain)
__global__ void k_readlane_dep(int n, long long* cyc, double* out)
{
    double x = 1.0 + threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
        {
            const int lo = __builtin_amdgcn_readlane((int)__double2loint(x), u);
            const int hi = __builtin_amdgcn_readlane((int)__double2hiint(x), u);
            const double b = __hiloint2double(hi, lo);
            x = fma(x, 1e-9, b);
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = x;
}
// ds_bpermute broadcast of a double + dependent fma
__global__ void k_bpermute_dep(int n, long long* cyc, double* out)
{
    double x = 1.0 + threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
        {
            const int lo = __builtin_amdgcn_ds_bpermute(4 * u, (int)__double2loint(x));
            const int hi = __builtin_amdgcn_ds_bpermute(4 * u, (int)__double2hiint(x));
            const double b = __hiloint2double(hi, lo);
            x = fma(x, 1e-9, b);
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = x;
}
// LDS write then broadcast read (same address on all lanes) + dependent fma: one hop of a
// "column to LDS, everybody reads it back" scheme
__global__ void k_lds_roundtrip(int n, long long* cyc, double* out)
{
    __shared__ double buf[128];
    double x = 1.0 + threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
        {
            buf[threadIdx.x & 63] = x;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const double b = buf[u];
            x = fma(x, 1e-9, b);
            __builtin_amdgcn_wave_barrier();
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = x;
}
// dependent LDS read chain, unrolled
__global__ void k_lds_dep(int n, long long* cyc, int* out)
{
    __shared__ int a[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x)
        a[i] = (i * 7 + 1) & 1023;
    __syncthreads();
    int i = threadIdx.x;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
            i = a[i];
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
    {
        cyc[0] = t1 - t0;
        out[0] = i;
    }
}
// barriers, unrolled, with 1024 threads
__global__ void k_barriers(int n, long long* cyc)
{
    long long t0 = clock64();
    for (int i = 0; i < n; i++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
            __syncthreads();
    }
    if (threadIdx.x == 0)
        cyc[0] = clock64() - t0;
}
// dependent mfma chain and mfma + dependent valu
__global__ void k_mfma_dep(int n, long long* cyc, double* out)
{
    d4 acc = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
// 4x4x4 fp64 mfma (4 blocks) dependent chain
__global__ void k_mfma4_dep(int n, long long* cyc, double* out)
{
    double acc = 0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = acc;
}
// DPP row broadcast style: v_mov_dpp quad_perm / row_bcast equivalents through __shfl (ds_bpermute
// or dpp chosen by the compiler) : width-16 shuffle from lane u
__global__ void k_shfl16_dep(int n, long long* cyc, double* out)
{
    double x = 1.0 + threadIdx.x * 1e-6;
    long long t0 = clock64();
    for (int k = 0; k < n; k++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
            x = fma(x, 1e-9, __shfl(x, u, 16));
    }
    long long t1 = clock64();
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
    out[threadIdx.x] = x;
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    long long* dc;
    CK(hipMalloc(&dc, 64));
    double* dd;
    CK(hipMalloc(&dd, 8192 * 2));
    long long cyc;
#define RUN(label, kern, thr, iters, per, ...)                                                   \
    do                                                                                           \
    {                                                                                            \
        for (int rep = 0; rep < 2; rep++)                                                        \
        {                                                                                        \
            hipLaunchKernelGGL(kern, dim3(1), dim3(thr), 0, s, iters, dc, ##__VA_ARGS__);        \
            CK(hipStreamSynchronize(s));                                                         \
        }                                                                                        \
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));                                       \
        printf("%-58s %8.2f cycles\n", label, (double)cyc / ((double)(iters) * (per)));          \
    } while (0)
    RUN("dependent fp64 fma, unroll 1 (loop overhead incl.)", (k_fma_dep<1>), 64, 4000, 1, dd);
    RUN("dependent fp64 fma, unroll 16", (k_fma_dep<16>), 64, 1000, 16, dd);
    RUN("dependent fp64 fma, unroll 64", (k_fma_dep<64>), 64, 500, 64, dd);
    RUN("dependent fp64 mul, unroll 64", (k_mul_dep<64>), 64, 500, 64, dd);
    RUN("dependent fp64 fma, unroll 64, 1024 thr (4 waves/SIMD)", (k_fma_dep<64>), 1024, 500, 64, dd);
    RUN("2 independent fp64 fma chains (per fma)", (k_fma_ind<2>), 64, 500, 32, dd);
    RUN("4 independent fp64 fma chains (per fma)", (k_fma_ind<4>), 64, 500, 64, dd);
    RUN("8 independent fp64 fma chains (per fma)", (k_fma_ind<8>), 64, 500, 128, dd);
    RUN("8 independent fp64 fma chains, 256 thr (per fma)", (k_fma_ind<8>), 256, 500, 128, dd);
    RUN("8 independent fp64 fma chains, 1024 thr (per fma per wave)", (k_fma_ind<8>), 1024, 500, 128, dd);
    RUN("v_rsq_f64 + add, dependent", k_rsq_dep, 64, 500, 16, dd);
    RUN("rsq + coupled Newton sqrt|rsqrt + fma (one pivot), dependent", k_rsq_newton, 64, 500, 8, dd);
    RUN("readlane x2 + fma, dependent", k_readlane_dep, 64, 500, 16, dd);
    RUN("ds_bpermute x2 + fma, dependent", k_bpermute_dep, 64, 500, 16, dd);
    RUN("__shfl(width 16) + fma, dependent", k_shfl16_dep, 64, 500, 16, dd);
    RUN("LDS store + wave barrier + broadcast load + fma, dependent", k_lds_roundtrip, 64, 500, 16, dd);
    RUN("dependent LDS read, unroll 16", k_lds_dep, 64, 500, 16, (int*)dd);
    RUN("dependent mfma_f64_16x16x4, unroll 16", k_mfma_dep, 64, 500, 16, dd);
    RUN("dependent mfma_f64_4x4x4 (4 blocks), unroll 16", k_mfma4_dep, 64, 500, 16, dd);
    for (int thr : {64, 256, 1024})
    {
        for (int rep = 0; rep < 2; rep++)
        {
            hipLaunchKernelGGL(k_barriers, dim3(1), dim3(thr), 0, s, 500, dc);
            CK(hipStreamSynchronize(s));
        }
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
        printf("__syncthreads, %4d threads, unroll 16 %28s %8.2f cycles\n", thr, "", (double)cyc / 8000.0);
    }
    return 0;
}
