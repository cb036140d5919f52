#include <cmath>
// micro-benchmarks that calibrate the latency model used for the Cholesky kernels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_lds(int* p) { extern __shared__ double lds[]; if (threadIdx.x == 0) lds[0] = 1; __syncthreads(); if (p && lds[0] == 2) p[0] = 1; }
// dependent pointer chase by one lane
__global__ void k_chase(const int* next, int n, int* out, long long* cyc)
{
    int i = 0;
    long long t0 = clock64();
    for (int k = 0; k < n; k++) i = next[i];
    long long t1 = clock64();
    out[0] = i; cyc[0] = t1 - t0;
}
// N barriers
__global__ void k_barriers(int n, long long* cyc) { long long t0 = clock64(); for (int i = 0; i < n; i++) __syncthreads(); if (threadIdx.x == 0) cyc[0] = clock64() - t0; }
// dependent LDS chain
__global__ void k_ldschain(int n, long long* cyc, int* out) { __shared__ int a[1024]; for (int i = threadIdx.x; i < 1024; i += blockDim.x) a[i] = (i * 7 + 1) & 1023; __syncthreads(); int i = threadIdx.x; long long t0 = clock64(); for (int k = 0; k < n; k++) i = a[i]; long long t1 = clock64(); if (threadIdx.x == 0) { cyc[0] = t1 - t0; out[0] = i; } }
// dependent fp64 fma chain
__global__ void k_fmachain(int n, long long* cyc, double* out) { double x = threadIdx.x * 1e-3, y = 1.0000001; long long t0 = clock64(); for (int k = 0; k < n; k++) x = fma(x, y, 1e-9); long long t1 = clock64(); if (threadIdx.x == 0) cyc[0] = t1 - t0; out[threadIdx.x] = x; }
__global__ void k_sqrtchain(int n, long long* cyc, double* out) { double x = 2.0 + threadIdx.x; long long t0 = clock64(); for (int k = 0; k < n; k++) x = sqrt(x) + 1.5; long long t1 = clock64(); if (threadIdx.x == 0) cyc[0] = t1 - t0; out[threadIdx.x] = x; }
__global__ void k_divchain(int n, long long* cyc, double* out) { double x = 2.0 + threadIdx.x; long long t0 = clock64(); for (int k = 0; k < n; k++) x = 3.0 / x + 1.5; long long t1 = clock64(); if (threadIdx.x == 0) cyc[0] = t1 - t0; out[threadIdx.x] = x; }

typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma_dep(int n, long long* cyc, double* out) { d4 acc = {0, 0, 0, 0}; double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6; long long t0 = clock64(); for (int k = 0; k < n; k++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); long long t1 = clock64(); if (threadIdx.x == 0) cyc[0] = t1 - t0; out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3]; }
__global__ void k_mfma_ind4(int n, long long* cyc, double* out) { d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0; double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6; long long t0 = clock64(); for (int k = 0; k < n; k++) { a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a3, 0, 0, 0); } long long t1 = clock64(); if (threadIdx.x == 0) cyc[0] = t1 - t0; out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3]; }
__global__ void k_fma_ind8(int n, long long* cyc, double* out) { double x[8]; for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3 + i; double y = 1.0000001; long long t0 = clock64(); for (int k = 0; k < n; k++) { for (int i = 0; i < 8; i++) x[i] = fma(x[i], y, 1e-9); } long long t1 = clock64(); if (threadIdx.x == 0) cyc[0] = t1 - t0; double s = 0; for (int i = 0; i < 8; i++) s += x[i]; out[threadIdx.x] = s; }
__global__ void k_f32_dep(int n, long long* cyc, double* out) { float x = threadIdx.x * 1e-3f, y = 1.0000001f; long long t0 = clock64(); for (int k = 0; k < n; k++) x = fmaf(x, y, 1e-9f); long long t1 = clock64(); if (threadIdx.x == 0) cyc[0] = t1 - t0; out[threadIdx.x] = x; }
__global__ void k_wallclock(long long* cyc) { long long t0 = clock64(); long long w0 = wall_clock64(); long long w1 = w0; while (w1 - w0 < 100000) w1 = wall_clock64(); long long t1 = clock64(); if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; } }

__global__ void k_rsq_acc(const double* d, double* r0, double* r1, double* r2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d[i];
    double r = __builtin_amdgcn_rsq(x);
    r0[i] = r;
    { double t = x * r; double e = fma(-t, r, 1.0); r = fma(0.5 * r, e, r); }
    r1[i] = r;
    { double t = x * r; double e = fma(-t, r, 1.0); r = fma(0.5 * r, e, r); }
    r2[i] = r;
}

int main()
{
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    int* d; CK(hipMalloc(&d, 1 << 20));
    long long* dc; CK(hipMalloc(&dc, 64)); double* dd; CK(hipMalloc(&dd, 8192));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, s, d);
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        printf("1000 empty kernels (1x256): %.2f us each\n", ms);
    }
    CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 140000));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(k_lds, dim3(8), dim3(1024), 130000, s, d);
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("1000 kernels 8x1024 thr, 130 KB LDS: %.2f us each\n", ms);
    CK(hipEventRecord(a, s));
    for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(k_lds, dim3(200), dim3(1024), 130000, s, d);
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("1000 kernels 200x1024 thr, 130 KB LDS: %.2f us each\n", ms);
    // pointer chase over a 256 MB buffer (stride ~ 1 MB + 64 B => every hop a new page / line)
    const size_t N = 64 << 20; int* big; CK(hipMalloc(&big, N * 4));
    std::vector<int> h(N, 0);
    size_t idx = 0; const int hops = 2000; const size_t stride = (1 << 18) + 16;
    for (int k = 0; k < hops; k++) { size_t nx = (idx + stride) % N; h[idx] = (int)nx; idx = nx; }
    CK(hipMemcpy(big, h.data(), N * 4, hipMemcpyHostToDevice));
    long long cyc;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, s, big, hops, d, dc); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("global pointer chase (1 MB stride, rep %d): %.0f cycles/hop (clock64 = 100 MHz ticks? see fma)\n", rep, (double)cyc / hops);
    }
    // small-stride chase (L2 hits)
    for (size_t i = 0; i < 4096; i++) h[i] = (int)((i * 67 + 1) % 4096);
    CK(hipMemcpy(big, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, s, big, 4000, d, dc); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("global pointer chase (16 KB footprint, rep %d): %.0f ticks/hop\n", rep, (double)cyc / 4000);
    }
    for (int thr : {64, 256, 1024}) {
        hipLaunchKernelGGL(k_barriers, dim3(1), dim3(thr), 0, s, 1000, dc); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("__syncthreads with %d threads: %.1f ticks each\n", thr, (double)cyc / 1000);
    }
    hipLaunchKernelGGL(k_ldschain, dim3(1), dim3(64), 0, s, 1000, dc, d); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("dependent LDS read: %.1f ticks\n", (double)cyc / 1000);
    hipLaunchKernelGGL(k_fmachain, dim3(1), dim3(64), 0, s, 10000, dc, dd); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("dependent fp64 fma: %.2f ticks\n", (double)cyc / 10000);
    hipLaunchKernelGGL(k_sqrtchain, dim3(1), dim3(64), 0, s, 2000, dc, dd); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("dependent fp64 sqrt+add: %.1f ticks\n", (double)cyc / 2000);
    hipLaunchKernelGGL(k_divchain, dim3(1), dim3(64), 0, s, 2000, dc, dd); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("dependent fp64 div+add: %.1f ticks\n", (double)cyc / 2000);
    for (int thr : {64, 256, 1024}) {
        hipLaunchKernelGGL(k_mfma_dep, dim3(1), dim3(thr), 0, s, 2000, dc, dd); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("dependent mfma_f64_16x16x4 (%d thr): %.1f ticks each\n", thr, (double)cyc / 2000);
        hipLaunchKernelGGL(k_mfma_ind4, dim3(1), dim3(thr), 0, s, 2000, dc, dd); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("4 independent mfma_f64_16x16x4 (%d thr): %.1f ticks per mfma\n", thr, (double)cyc / 8000);
        hipLaunchKernelGGL(k_fma_ind8, dim3(1), dim3(thr), 0, s, 2000, dc, dd); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("8 independent fp64 fma chains (%d thr): %.2f ticks per fma\n", thr, (double)cyc / 16000);
    }
    hipLaunchKernelGGL(k_f32_dep, dim3(1), dim3(64), 0, s, 10000, dc, dd); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("dependent fp32 fma: %.2f ticks\n", (double)cyc / 10000);
    { long long c2[2]; hipLaunchKernelGGL(k_wallclock, dim3(1), dim3(64), 0, s, dc); CK(hipStreamSynchronize(s));
      CK(hipMemcpy(c2, dc, 16, hipMemcpyDeviceToHost)); printf("clock64 ticks per wall_clock64 tick (100 MHz): %.3f => clock64 at %.1f MHz\n", (double)c2[0] / c2[1], 100.0 * c2[0] / c2[1]); }
    { // accuracy of v_rsq_f64 and of one / two Newton steps on it
        const int n = 1 << 20; std::vector<double> hd(n), h0(n), h1(n), h2(n);
        unsigned long long st = 88172645463325252ULL;
        for (int i = 0; i < n; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; double u = (st >> 11) * (1.0 / 9007199254740992.0); hd[i] = exp(-7.0 + 21.0 * u); }
        double *dx, *d0, *d1, *d2; CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&d0, n * 8)); CK(hipMalloc(&d1, n * 8)); CK(hipMalloc(&d2, n * 8));
        CK(hipMemcpy(dx, hd.data(), n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_rsq_acc, dim3(n / 256), dim3(256), 0, s, dx, d0, d1, d2, n); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h0.data(), d0, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), d1, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), d2, n * 8, hipMemcpyDeviceToHost));
        double e0 = 0, e1 = 0, e2 = 0;
        for (int i = 0; i < n; i++) { long double ref = 1.0L / sqrtl((long double)hd[i]); e0 = fmax(e0, (double)fabsl((h0[i] - ref) / ref)); e1 = fmax(e1, (double)fabsl((h1[i] - ref) / ref)); e2 = fmax(e2, (double)fabsl((h2[i] - ref) / ref)); }
        printf("v_rsq_f64 max rel err %.3e ; +1 Newton %.3e ; +2 Newton %.3e (eps = 2.2e-16)\n", e0, e1, e2);
    }
    // wall time of the fma kernel to convert ticks -> ns
    CK(hipEventRecord(a, s));
    hipLaunchKernelGGL(k_fmachain, dim3(1), dim3(64), 0, s, 2000000, dc, dd);
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); printf("2M fma chain: %.3f ms wall, %lld ticks => %.3f ticks/ns ; %.2f ns per dependent fma\n", ms, cyc, cyc / (ms * 1e6), ms * 1e6 / 2e6);
    return 0;
}
